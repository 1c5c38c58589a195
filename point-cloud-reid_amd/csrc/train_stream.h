// launch-side interface of train_stream_kernels.hip (wave-autonomous train-dense kernels for the narrow grouped-MLP layers)
#pragma once
#include "pcr_common.h"

bool pcr_ts_fwd_ok(const pcr_tdense_fwd *p);
bool pcr_ts_fwd_pools(const pcr_tdense_fwd *p);             // the launch fills pool_ymax / pool_arg
int pcr_ts_fwd_grid(const pcr_tdense_fwd *p, int *per);      // workgroups = rows of the launch's statistics partials
int pcr_ts_fwd_launch(const pcr_tdense_fwd *p, hipStream_t st);
bool pcr_ts_bwd_ok(const pcr_tdense_bwd *p);
int pcr_ts_bwd_grid(const pcr_tdense_bwd *p, int *per);      // workgroups = rows of the launch's dW / db / dstats partials
int pcr_ts_bwd_launch(const pcr_tdense_bwd *p, hipStream_t st);
