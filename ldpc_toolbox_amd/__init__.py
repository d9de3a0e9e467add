"""MI355X-native batched LDPC belief-propagation decoder behind the ldpc-toolbox decoder
boundary.  The product is libldpc_toolbox.so (HIP kernels + C ABI, see include/ldpc_toolbox.h);
this package is the Python mirror of the reference's interface for that path."""
from . import _capi
from .decoder import (ALL_IMPLEMENTATIONS, FAST_IMPLEMENTATIONS, DecoderImplementation, DecoderOutput, DecoderUnavailable, Encoder,
                      I8_IMPLEMENTATIONS, IMPLEMENTATIONS, LdpcDecoder, Simulator)
from .sparse import SparseMatrix

code_alist = _capi.code_alist

__all__ = ["ALL_IMPLEMENTATIONS", "FAST_IMPLEMENTATIONS", "I8_IMPLEMENTATIONS", "DecoderImplementation", "DecoderOutput", "DecoderUnavailable", "Encoder",
           "IMPLEMENTATIONS", "LdpcDecoder", "Simulator", "SparseMatrix", "code_alist"]
