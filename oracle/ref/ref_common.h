// oracle/ref -- TEST INFRASTRUCTURE: a harness around the REFERENCE's own object code.
//
// The reference (naxingyu/lstm-rnn, /root/reference) cannot be built as a whole in this image: TrainableLayer.cu,
// InputLayer.cpp, PostOutputLayer.cpp and everything above them need Boost, which is absent, and stand-ins for missing
// libraries are not permitted.  What DOES compile from the reference's own sources, unmodified and where they lie, is the
// arithmetic of the hot path:
//     helpers/Matrix.cu                         (the Cpu products, Matrix.cu:41-183,218-349)      -> its own object
//     layers/LstmLayer.cu                       (ComputeBlockOutputFn, ComputeBlockErrorsFn, ComputeWeightUpdateFn, Resort*)
//     layers/FeedForwardLayer.cu                (ComputeOutputFn, ComputeDeltaFn, ComputeBiasWeightUpdateFn)
//     layers/SoftmaxLayer.cu                    (CalculateOffsetFn ... CalculateErrorsFn)
//     layers/MulticlassClassificationLayer.cu   (ComputeCrossEntropyErrorFn, CountCorrectClassificationsFn, ComputeOutputErrorFn)
// The functors live in anonymous namespaces, so each ref_*.cpp here #includes ONE reference .cu by absolute path and drives
// its functors and helpers::Matrix<Cpu> in the order of the reference's computeForwardPass / computeBackwardPass (cited
// per block).  The layer CLASSES of those files are compiled too but never constructed (their base class lives in the
// unbuildable TrainableLayer.cu); the link drops them (-ffunction-sections, --gc-sections, hidden visibility).
// hipcc is used as a host compiler only (rocThrust's CPP backend: THRUST_DEVICE_SYSTEM_CPP; g++ cannot parse rocPRIM).
//
// What this pins: every arithmetic statement and summation order of the path (functors + Cpu GEMM).  What it does not:
// the ~40 lines of time-loop orchestration per pass are restated here from LstmLayer.cu:763-1051 (they are call sequences
// only: which functor, which time step, which flags).  KAT-0 (tests/test_oracle_kat0.py) pins those end to end.
//
// Output goes to oracle/_ref/ only (git-ignored); nothing in lstm-rnn_amd/ or bench.py's timed region touches it.
#pragma once
#include <cstring>
#define REF_API extern "C" __attribute__((visibility("default")))
