// Drives the functors of the reference's remaining post output layers (see ref_common.h).  ONE source, compiled once per
// layer file with -DREF_POST=<kind>: every one of those .cu files defines `internal::(anonymous)::ComputeOutputErrorFn`,
// so each needs a translation unit of its own.  Kinds = POST_* of oracle/currennt_oracle.c:
//     0 sse          layers/SsePostOutputLayer.cu          ComputeSseFn :39-60, ComputeOutputErrorFn :62-88
//     1 weightedsse  layers/WeightedSsePostOutputLayer.cu  ComputeWeightedSseFn :40-64, ComputeOutputErrorFn :66-92
//     2 wf           layers/SseMaskPostOutputLayer.cu      ComputeSseMaskFn :40-64, ComputeOutputErrorFn :66-92
//     3 ce           layers/CePostOutputLayer.cu           ComputeCeFn :43-71, ComputeOutputErrorFn :73-99
//     4 rmse         layers/RmsePostOutputLayer.cu         ComputeRmseFn :40-71, ComputeOutputErrorFn :73-96
//     5 binary       layers/BinaryClassificationLayer.cu   ComputeCrossEntropyErrorFn :44-67, CountCorrect... :69-85,
//                                                          ComputeOutputErrorFn :87-111
// The thrust calls below are the ones of each layer's calculateError / computeForwardPass / computeBackwardPass with the
// member accessors replaced by local vectors.  L = size of the OUTPUT layer (kinds 1, 2: the post layer's size() / 2).
#include <thrust/transform_reduce.h>
#include <thrust/transform.h>
#include <thrust/reduce.h>
#include <thrust/for_each.h>
#if REF_POST == 0
#include "/root/reference/currennt_lib/src/layers/SsePostOutputLayer.cu"
#define KIND_NAME sse
#elif REF_POST == 1
#include "/root/reference/currennt_lib/src/layers/WeightedSsePostOutputLayer.cu"
#define KIND_NAME weightedsse
#elif REF_POST == 2
#include "/root/reference/currennt_lib/src/layers/SseMaskPostOutputLayer.cu"
#define KIND_NAME wf
#elif REF_POST == 3
#include "/root/reference/currennt_lib/src/layers/CePostOutputLayer.cu"
#define KIND_NAME ce
#elif REF_POST == 4
#include "/root/reference/currennt_lib/src/layers/RmsePostOutputLayer.cu"
#define KIND_NAME rmse
#elif REF_POST == 5
#include "/root/reference/currennt_lib/src/layers/BinaryClassificationLayer.cu"
#define KIND_NAME binary
#else
#error "REF_POST must be 0..5"
#endif
#include "ref_common.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define HIDDEN extern "C" __attribute__((visibility("hidden")))

namespace {
struct Vectors {                       // what PostOutputLayer<Cpu> holds (PostOutputLayer.hpp:45-47, Layer.hpp:61)
    Cpu::pattype_vector patTypes;
    Cpu::real_vector targets, actualOutputs, outputErrors;
    Vectors(int nOut, int nTgt, int N, const char *pt, const real_t *t, const real_t *y, const real_t *e)
        : patTypes(pt, pt + N), targets(t, t + nTgt), actualOutputs(y, y + nOut), outputErrors(nOut)
    {
        if (e) thrust::copy(e, e + nOut, outputErrors.begin());
        else thrust::fill(outputErrors.begin(), outputErrors.end(), (real_t)0);
    }
};
}

#if REF_POST == 0 || REF_POST == 3
// SsePostOutputLayer<Cpu>::calculateError .cu:114-132 / CePostOutputLayer<Cpu>::calculateError .cu:125-143
HIDDEN real_t CAT(ref_post_error_, KIND_NAME)(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    int n = N * L;
    Vectors v(n, n, N, patTypes, targets, outputs, 0);
#if REF_POST == 0
    internal::ComputeSseFn fn;
#else
    internal::ComputeCeFn fn;
#endif
    fn.layerSize = L;
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    real_t r = thrust::transform_reduce(
        thrust::make_zip_iterator(thrust::make_tuple(v.targets.begin(), v.actualOutputs.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(v.targets.begin() + n, v.actualOutputs.begin() + n, thrust::counting_iterator<int>(0) + n)),
        fn, (real_t)0, thrust::plus<real_t>());
#if REF_POST == 0
    return (real_t)0.5 * r;
#else
    return r;
#endif
}

// SsePostOutputLayer<Cpu>::computeBackwardPass .cu:139-155 / CePostOutputLayer<Cpu>::computeBackwardPass .cu:150-170
HIDDEN void CAT(ref_post_backward_, KIND_NAME)(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs, real_t *outErr)
{
    int n = N * L;
    Vectors v(n, n, N, patTypes, targets, outputs, outErr);
    internal::ComputeOutputErrorFn fn;
    fn.layerSize = L;
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    thrust::transform(
        thrust::make_zip_iterator(thrust::make_tuple(v.actualOutputs.begin(), v.targets.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(v.actualOutputs.begin() + n, v.targets.begin() + n, thrust::counting_iterator<int>(0) + n)),
        v.outputErrors.begin(), fn);
    memcpy(outErr, helpers::getRawPointer(v.outputErrors), (size_t)n * sizeof(real_t));
}
#endif

#if REF_POST == 1 || REF_POST == 2
// WeightedSsePostOutputLayer<Cpu>::calculateError .cu:119-139 / SseMaskPostOutputLayer<Cpu>::calculateError .cu:119-139
HIDDEN real_t CAT(ref_post_error_, KIND_NAME)(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    int n = N * L;                      // = curMaxSeqLength * parallelSequences * size() / 2
    Vectors v(n, 2 * n, N, patTypes, targets, outputs, 0);
#if REF_POST == 1
    internal::ComputeWeightedSseFn fn;
#else
    internal::ComputeSseMaskFn fn;
#endif
    fn.layerSize = L;
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    fn.targets = helpers::getRawPointer(v.targets);
    fn.outputs = helpers::getRawPointer(v.actualOutputs);
    return (real_t)0.5 * thrust::transform_reduce(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + n,
                                                  fn, (real_t)0, thrust::plus<real_t>());
}

// ...::computeBackwardPass .cu:146-167 (both files)
HIDDEN void CAT(ref_post_backward_, KIND_NAME)(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs, real_t *outErr)
{
    int n = N * L;
    Vectors v(n, 2 * n, N, patTypes, targets, outputs, outErr);
    internal::ComputeOutputErrorFn fn;
    fn.layerSize = L;
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    fn.targets = helpers::getRawPointer(v.targets);
    fn.outputs = helpers::getRawPointer(v.actualOutputs);
    thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + n, v.outputErrors.begin(), fn);
    memcpy(outErr, helpers::getRawPointer(v.outputErrors), (size_t)n * sizeof(real_t));
}
#endif

#if REF_POST == 4
// RmsePostOutputLayer<Cpu>::computeForwardPass .cu:139-152 (per-pattern RMSEs), then calculateError .cu:125-133
static void rmse_forward(Vectors &v, Cpu::real_vector &m_rmses, int L, int N)
{
    internal::ComputeRmseFn fn;
    fn.layerSize = L;
    fn.actualOutputs = helpers::getRawPointer(v.actualOutputs);
    fn.targetOutputs = helpers::getRawPointer(v.targets);
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    thrust::transform(thrust::counting_iterator<int>(0), thrust::counting_iterator<int>(0) + N, m_rmses.begin(), fn);
}

HIDDEN real_t ref_post_error_rmse(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    Vectors v(N * L, N * L, N, patTypes, targets, outputs, 0);
    Cpu::real_vector m_rmses(N);
    rmse_forward(v, m_rmses, L, N);
    return thrust::reduce(m_rmses.begin(), m_rmses.begin() + N);
}

// RmsePostOutputLayer<Cpu>::computeBackwardPass .cu:154-174
HIDDEN void ref_post_backward_rmse(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs, real_t *outErr)
{
    int n = N * L;
    Vectors v(n, n, N, patTypes, targets, outputs, outErr);
    Cpu::real_vector m_rmses(N);
    rmse_forward(v, m_rmses, L, N);
    internal::ComputeOutputErrorFn fn;
    fn.layerSize = L;
    fn.rmses = helpers::getRawPointer(m_rmses);
    thrust::transform(
        thrust::make_zip_iterator(thrust::make_tuple(v.actualOutputs.begin(), v.targets.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(v.actualOutputs.begin() + n, v.targets.begin() + n, thrust::counting_iterator<int>(0) + n)),
        v.outputErrors.begin(), fn);
    memcpy(outErr, helpers::getRawPointer(v.outputErrors), (size_t)n * sizeof(real_t));
}
#endif

#if REF_POST == 5
// BinaryClassificationLayer<Cpu>::calculateError .cu:166-183 (L == 1; targets = the target classes as reals, .cu:157-164)
HIDDEN real_t ref_post_error_binary(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    (void)L;
    int n = N;
    Vectors v(n, n, N, patTypes, targets, outputs, 0);
    internal::ComputeCrossEntropyErrorFn fn;
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    return thrust::transform_reduce(
        thrust::make_zip_iterator(thrust::make_tuple(v.targets.begin(), v.actualOutputs.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(v.targets.begin() + n, v.actualOutputs.begin() + n, thrust::counting_iterator<int>(0) + n)),
        fn, (real_t)0, thrust::plus<real_t>());
}

// BinaryClassificationLayer<Cpu>::countCorrectClassifications .cu:132-148
REF_API int ref_binary_correct(int N, const char *patTypes, const real_t *targets, const real_t *outputs)
{
    int n = N;
    Vectors v(n, n, N, patTypes, targets, outputs, 0);
    internal::CountCorrectClassificationsFn fn;
    return thrust::transform_reduce(
        thrust::make_zip_iterator(thrust::make_tuple(v.targets.begin(), v.actualOutputs.begin(), v.patTypes.begin())),
        thrust::make_zip_iterator(thrust::make_tuple(v.targets.begin() + n, v.actualOutputs.begin() + n, v.patTypes.begin() + n)),
        fn, 0, thrust::plus<int>());
}

// BinaryClassificationLayer<Cpu>::computeBackwardPass .cu:190-207: dummy slots keep what outputErrors held before
HIDDEN void ref_post_backward_binary(int L, int N, const char *patTypes, const real_t *targets, const real_t *outputs, real_t *outErr)
{
    (void)L;
    int n = N;
    Vectors v(n, n, N, patTypes, targets, outputs, outErr);
    internal::ComputeOutputErrorFn fn;
    fn.patTypes = helpers::getRawPointer(v.patTypes);
    thrust::for_each(
        thrust::make_zip_iterator(thrust::make_tuple(v.outputErrors.begin(), v.targets.begin(), v.actualOutputs.begin(), thrust::counting_iterator<int>(0))),
        thrust::make_zip_iterator(thrust::make_tuple(v.outputErrors.begin() + n, v.targets.begin() + n, v.actualOutputs.begin() + n, thrust::counting_iterator<int>(0) + n)),
        fn);
    memcpy(outErr, helpers::getRawPointer(v.outputErrors), (size_t)n * sizeof(real_t));
}
#endif
