// Grouped set-abstraction MLP, layers 2 / 3 in plain bf16 with f32 accumulation (body: sa_kernels_impl.h).
#define PCR_SA_PREC 2
#include "sa_kernels_impl.h"
