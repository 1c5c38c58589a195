// Grouped set-abstraction MLP, layers 2 / 3 as split bf16 on v_mfma_f32_32x32x16_bf16 (body: sa_kernels_impl.h).
#define PCR_SA_PREC 1
#include "sa_kernels_impl.h"
