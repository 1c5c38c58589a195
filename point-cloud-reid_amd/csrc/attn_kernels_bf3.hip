// Linear attention with its dense phases as split bf16 on v_mfma_f32_32x32x16_bf16 (body: attn_kernels_impl.h).
#define PCR_ATTN_PREC 1
#include "attn_kernels_impl.h"
