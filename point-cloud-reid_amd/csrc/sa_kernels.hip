// Grouped set-abstraction MLP, f32-input MFMA form + the C-ABI entry points (body: sa_kernels_impl.h).
#define PCR_SA_PREC 0
#include "sa_kernels_impl.h"
