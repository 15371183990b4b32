"""lstm-rnn_amd: MI355X (gfx950) implementation of the CURRENNT LSTM training hot path.

The product is the C-ABI library `libcurrennt_hip.so` (sources in csrc/, header in
/include/currennt_hip.h).  This package holds the ctypes binding of that ABI and a thin host-side
mirror of the reference's NeuralNetwork / Layer interface (currennt_lib/src/NeuralNetwork.cpp,
layers/Layer.hpp) used by the tests and bench.py.  The directory name contains a hyphen, so it is
loaded through `__graft_entry__.load_package()` under the module name `lstm_rnn_amd`.
"""
from .binding import (CurrenntHipError, lib_path, load_library, build_library,  # noqa: F401
                      PREC_F32, PREC_BF16, PREC_BF16X3, LAYER_KINDS, BUF)
from .fraction import make_fraction, make_fractions, PATTYPE_NONE  # noqa: F401
from .network import NeuralNetwork  # noqa: F401
from . import parallel  # noqa: F401
