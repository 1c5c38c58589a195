// Linear attention, f32-input MFMA form + the C-ABI entry points (body: attn_kernels_impl.h).
#define PCR_ATTN_PREC 0
#include "attn_kernels_impl.h"
