// Linear-attention kernels (pcr_attn_kv_f32 / pcr_attn_apply_f32).
#include "tile_dense.h"

namespace {
// -------------------------------------------------------------- linear attention ----
struct AttnArgs {
  pcr_attn_params p;
};

// Algebra used by both kernels (the host folds it into the weights, see AttnPlan in
// pcr_amd/engine.py): with h = relu(W0 xyz + b0) the position encoding is W2 h + b2, so
//   Wq (x + W2 h + b2) = [Wq | Wq W2] [x ; h] + Wq b2        (one dense instead of three)
//   [K ; V] pre-activation = [[Wk | kpos Wk W2] ; [Wv | Wv W2]] [x ; h] + [kpos Wk b2 ; Wv b2]
// and the merge projection is folded into the per-cloud KV matrix by the kv kernel:
//   merge(msg)[o] = sum_dd M[o][dd] Q'[dd],  M[o][dd] = sum_{v in head(dd)} Wm[o][v] KV[dd][v],
//   Q'[dd] = Q[dd] * Sk / (Q_head . ksum_head + 1e-6).

// hidden = relu(W0 xyz + b0) for the T tokens of a tile -> dst rows [0,d) ([d][RP]); zero xyz beyond L
__device__ __forceinline__ void pos_hidden(float *dst, int RP, const float *P, const float *w0,
                                           const float *b0, int d, int T) {
  for (int e = threadIdx.x; e < d * T; e += blockDim.x) {
    const int o = e / T, t = e - o * T;
    const float v = w0[o * 3] * P[t] + w0[o * 3 + 1] * P[RP + t] + w0[o * 3 + 2] * P[2 * RP + t] + b0[o];
    dst[o * RP + t] = fmaxf(v, 0.f);
  }
}

__device__ __forceinline__ void load_xyz3(float *P, int RP, const float *xyz, int L, int t0, int T) {
  for (int e = threadIdx.x; e < 3 * T; e += blockDim.x) {
    const int c = e / T, t = e - c * T;
    P[c * RP + t] = t0 + t < L ? xyz[(size_t)(t0 + t) * 3 + c] : 0.f;
  }
}

// One workgroup per key-side cloud, token tiles of T = 32*TB (TB = 2 for d = 32, else 1: the fused
// K/V projection has 2d/32 >= 4 cout blocks, one per wave).
// kv image per cloud: packed (d x d) matrix M (merge folded in, see above) followed by ksum[d].
// LDS: XH [c2 + d] key features ; hidden, KB [d], VB [d], P [3]; after the loop KVl [d][d+1].
template <int TB, int NR>
__global__ __launch_bounds__(kThreads) void attn_kv_kernel(AttnArgs a) {
  constexpr int T = 32 * TB, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c2 = p.c2;
  float *XH = smem;
  float *KB = XH + (c2 + d) * RP;
  float *VB = KB + d * RP;
  float *P = VB + d * RP;
  float *s_w0 = P + 3 * RP, *s_b0 = s_w0 + 3 * d;   // staged pos-MLP first layer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const size_t b = blockIdx.x;
  for (int e = tid; e < 3 * d; e += kThreads) s_w0[e] = p.pos0_w[e];
  for (int e = tid; e < d; e += kThreads) s_b0[e] = p.pos0_b[e];
  const float *feat = p.feat_k + b * c2 * p.Sk;
  const float *xyz = p.xyz_k + b * p.Sk * 3;
  const int nb = d >> 5, nT = nb * nb;
  const int dh = d / p.nhead;
  const float sk = (float)p.Sk;
  const float *bkv = p.bkv;

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  float ksum = 0.f;

  for (int t0 = 0; t0 < p.Sk; t0 += T) {
    const int valid = p.Sk - t0;
    load_tile(XH, RP, feat, c2, c2, p.Sk, t0, T);
    load_xyz3(P, RP, xyz, p.Sk, t0, T);
    __syncthreads();
    pos_hidden(XH + c2 * RP, RP, P, s_w0, s_b0, d, T);
    __syncthreads();
    tile_dense2<TB, NR>(XH, c2 + d, p.wkv, 2 * d, false, [&](float v, int o, int t) {
      if (o < d) KB[o * RP + t] = t < valid ? elu1(v) : 0.f;
      else VB[(o - d) * RP + t] = t < valid ? v / sk : 0.f;
    }, bkv);   // biases seed the accumulators
    __syncthreads();
    if (tid < d) {
      const float *row = KB + tid * RP;
      float s = 0.f;
      for (int t = 0; t < T; t++) s += row[t];
      ksum += s;
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int item = wave + 4 * it;
      if (item < nT) {
        const int ib = item / nb, jb = item - ib * nb;
        const float *ap = KB + (ib * 32 + l31) * RP + h;
        const float *bp = VB + (jb * 32 + l31) * RP + h;
#pragma unroll 4
        for (int ks = 0; ks < T / 2; ks++)
          acc[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[it], 0, 0, 0);
      }
    }
    // no barrier here: the next tile only rewrites XH/P before its first barrier, KB/VB after it
  }
  __syncthreads();
  // KV (head-masked) -> LDS [dd][d+1], then fold the merge projection and write the packed image
  float *KVl = smem;
  const int ld = d + 1;
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const int item = wave + 4 * it;
    if (item < nT) {
      const int ib = item / nb, jb = item - ib * nb;
      const int v = jb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int dd = ib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        KVl[dd * ld + v] = (dd / dh == v / dh) ? acc[it][r] : 0.f;
      }
    }
  }
  __syncthreads();
  float *kv = p.kv + b * ((size_t)d * d + d);
  for (int e = tid; e < d * d; e += kThreads) {
    const int o = e / d, dd = e - o * d;
    const int v0 = (dd / dh) * dh;
    const float *wm = p.wmerge + (size_t)o * d + v0;
    const float *kr = KVl + dd * ld + v0;
    float m = 0.f;
#pragma unroll 8
    for (int v = 0; v < dh; v++) m += wm[v] * kr[v];   // (unrolled: batches of independent loads)
    const int kb = dd >> 3, rem = dd & 7;
    kv[(((size_t)kb * d + o) * 2 + (rem & 1)) * 4 + (rem >> 1)] = m;
  }
  if (tid < d) kv[(size_t)d * d + tid] = ksum;
}

// One workgroup per (query cloud, tile of T query tokens), T = 128 / 64 / 32 for d = 32 / 64 / 128.
// LDS: CAT [c1 + d (pad 8)]: rows [0,c1) query features, rows [c1,c1+d) position hidden -> later
// the merged message; W [max(2d,cout,cfinal)] working buffer; P [3]; zs [nhead]; red.
template <int TB, int NR>
__global__ __launch_bounds__(kThreads) void attn_apply_kernel(AttnArgs a) {
  constexpr int T = 32 * TB, RP = T + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const pcr_attn_params &p = a.p;
  const int d = p.d, c1 = p.c1, cout = p.cout;
  const int catC = c1 + d, catP = ceil8(catC);
  int rowsW = 2 * d;
  if (ceil32(cout) > rowsW) rowsW = ceil32(cout);
  if (ceil32(p.cfinal) > rowsW) rowsW = ceil32(p.cfinal);
  float *CAT = smem;
  float *W = CAT + catP * RP;
  float *P = W + rowsW * RP;
  float *zs = P + 3 * RP;
  float *red = zs + p.nhead * RP;  // [2 * (256/T)][T]
  // small constant vectors, staged once: reading them from global inside the per-element loops costs a
  // vector-memory instruction per use
  float *cst = red + 2 * (kThreads / T) * T;
  float *s_ksum = cst, *s_ln1g = cst + d, *s_ln1b = cst + 2 * d, *s_ln2g = cst + 3 * d, *s_ln2b = s_ln2g + cout;
  float *s_w0 = s_ln2b + cout, *s_b0 = s_w0 + 3 * d;
  const int tid = threadIdx.x;
  const size_t b = blockIdx.y;
  const int t0 = blockIdx.x * T;
  const size_t bq_ = p.q_index ? (size_t)p.q_index[b] : b;    // which cloud supplies the query tokens
  const float *feat = p.feat_q + bq_ * c1 * p.Lq;
  const size_t kb_ = p.kv_index ? (size_t)p.kv_index[b] : b;
  const float *kv = p.kv + kb_ * ((size_t)d * d + d);
  const int dh = d / p.nhead;
  for (int e = tid; e < d; e += kThreads) {
    s_ksum[e] = kv[(size_t)d * d + e];
    s_ln1g[e] = p.ln1_g[e];
    s_ln1b[e] = p.ln1_b[e];
    if (p.q_pos) {
      s_w0[3 * e] = p.pos0_w[3 * e];
      s_w0[3 * e + 1] = p.pos0_w[3 * e + 1];
      s_w0[3 * e + 2] = p.pos0_w[3 * e + 2];
      s_b0[e] = p.pos0_b[e];
    }
  }
  for (int e = tid; e < cout; e += kThreads) {
    s_ln2g[e] = p.ln2_g[e];
    s_ln2b[e] = p.ln2_b[e];
  }
  const float *ksum = s_ksum;

  load_tile(CAT, RP, feat, c1, c1, p.Lq, t0, T);
  if (p.q_pos) {
    load_xyz3(P, RP, p.xyz_q + bq_ * p.Lq * 3, p.Lq, t0, T);
    __syncthreads();
    pos_hidden(CAT + c1 * RP, RP, P, s_w0, s_b0, d, T);
    for (int e = tid; e < (catP - catC) * T; e += kThreads) CAT[(catC + e / T) * RP + e % T] = 0.f;
  } else {
    for (int e = tid; e < (catP - c1) * T; e += kThreads) CAT[(c1 + e / T) * RP + e % T] = 0.f;
  }
  __syncthreads();
  {  // Q = elu(Wq' [x ; h] + bq) + 1
    tile_dense2<TB, NR>(CAT, p.q_pos ? catP : ceil8(c1), p.wq, d, false,
                       [&](float v, int o, int t) { W[o * RP + t] = elu1(v); }, p.bq);
  }
  __syncthreads();
  for (int e = tid; e < p.nhead * T; e += kThreads) {
    const int hd = e / T, t = e - hd * T;
    float z = 0.f;
    for (int c = 0; c < dh; c++) z += W[(hd * dh + c) * RP + t] * ksum[hd * dh + c];
    zs[hd * RP + t] = (1.0f / (z + 1e-6f)) * (float)p.Sk;
  }
  __syncthreads();
  for (int e = tid; e < d * T; e += kThreads) {
    const int o = e / T, t = e - o * T;
    W[o * RP + t] *= zs[(o / dh) * RP + t];
  }
  __syncthreads();
  tile_dense2<TB, NR>(W, d, kv, d, false, [&](float v, int o, int t) { CAT[(c1 + o) * RP + t] = v; });
  __syncthreads();
  tile_layernorm(CAT + c1 * RP, d, RP, T, s_ln1g, s_ln1b, red);
  tile_dense2<TB, NR>(CAT, catP, p.wmlp0, 2 * d, false, [&](float v, int o, int t) { W[o * RP + t] = fmaxf(v, 0.f); });
  __syncthreads();
  tile_dense2<TB, NR>(W, 2 * d, p.wmlp2, ceil32(cout), true, [&](float v, int o, int t) { W[o * RP + t] = v; });
  __syncthreads();
  tile_layernorm(W, cout, RP, T, s_ln2g, s_ln2b, red);
  if (p.residual) {
    for (int e = tid; e < cout * T; e += kThreads) {
      const int c = e / T, t = e - c * T;
      W[c * RP + t] = CAT[c * RP + t] + W[c * RP + t];
    }
    __syncthreads();
  }
  int cres = cout;
  if (p.cfinal) {  // trailing 1x1 conv with bias (cov_final); needs cout % 8 == 0
    const int cf = p.cfinal;
    tile_dense2<TB, NR>(W, cout, p.wfinal, ceil32(cf), true, [&](float v, int o, int t) { W[o * RP + t] = v; },
                        p.bfinal);   // bfinal is zero-padded to a multiple of 32 by the host
    __syncthreads();
    cres = cf;
  }
  float *out = p.out + b * cres * p.Lq;
  for (int e = tid; e < cres * T; e += kThreads) {
    const int c = e / T, t = e - c * T;
    if (t0 + t < p.Lq) out[(size_t)c * p.Lq + t0 + t] = W[c * RP + t];
  }
}

}  // namespace

static int attn_check(const pcr_attn_params &p) {
  if (p.B < 0 || p.Lq < 1 || p.Sk < 1 || p.c1 < 1 || p.c2 < 1 || p.cout < 1 || p.nhead < 1) return 1;
  if (p.d < 32 || p.d > 128 || (p.d & 31) || p.d % p.nhead) return 1;  // d_model in {32,64,96,128}
  if ((p.c2 & 7) || p.cout > 256 || p.cfinal > 256) return 1;
  if (!p.feat_q || !p.feat_k || !p.xyz_k || !p.kv || !p.pos0_w || !p.pos0_b || !p.wq || !p.bq || !p.wkv ||
      !p.bkv || !p.wmerge || !p.wmlp0 || !p.wmlp2 || !p.ln1_g || !p.ln1_b || !p.ln2_g || !p.ln2_b)
    return 1;
  if (p.q_pos && (!p.xyz_q || p.c1 != p.c2 || p.c1 != p.d)) return 1;
  if (p.residual && p.cout != p.c1) return 1;
  if (p.cfinal && (!p.wfinal || !p.bfinal || (p.cout & 7))) return 1;
  return 0;
}

PCR_EXPORT int pcr_attn_kv_f32(const pcr_attn_params *pp, pcr_stream_t stream) {
  if (!pp || attn_check(*pp)) return PCR_ERR_INVALID;
  if (pp->B == 0) return PCR_OK;
  AttnArgs a;
  a.p = *pp;
  const int d = pp->d;
  const int tb = d <= 32 ? 2 : 1, RP = 32 * tb + 1;
  size_t lds = ((size_t)(pp->c2 + 3 * d + 3) * RP + 4 * d) * sizeof(float);
  const size_t lds2 = (size_t)d * (d + 1) * sizeof(float);
  if (lds2 > lds) lds = lds2;
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(attn_kv_kernel<1, 1>) && allow_big_lds(attn_kv_kernel<1, 2>) &&
                   allow_big_lds(attn_kv_kernel<2, 1>);
  (void)ok;
  dim3 g(pp->B), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  // NR = 2 only when the fused K/V projection has more than four cout blocks (2d > 128)
  if (tb == 2) hipLaunchKernelGGL((attn_kv_kernel<2, 1>), g, blk, lds, st, a);
  else if (2 * d > 128) hipLaunchKernelGGL((attn_kv_kernel<1, 2>), g, blk, lds, st, a);
  else hipLaunchKernelGGL((attn_kv_kernel<1, 1>), g, blk, lds, st, a);
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

PCR_EXPORT int pcr_attn_apply_f32(const pcr_attn_params *pp, pcr_stream_t stream) {
  if (!pp || attn_check(*pp) || !pp->out) return PCR_ERR_INVALID;
  if (pp->B == 0) return PCR_OK;
  if (pp->B > 65535) return PCR_ERR_INVALID;
  const pcr_attn_params &p = *pp;
  AttnArgs a;
  a.p = p;
  const int tb = p.d <= 32 ? 4 : (p.d <= 64 ? 2 : 1), T = 32 * tb, RP = T + 1;
  const int catP = ceil8(p.c1 + p.d);
  int rowsW = 2 * p.d;
  if (ceil32(p.cout) > rowsW) rowsW = ceil32(p.cout);
  if (ceil32(p.cfinal) > rowsW) rowsW = ceil32(p.cfinal);
  size_t lds = ((size_t)(catP + rowsW + 3 + p.nhead) * RP + 2 * (kThreads / T) * T + 7 * p.d + 2 * p.cout) *
               sizeof(float);
  if (lds > (size_t)kMaxDynLds) return PCR_ERR_INVALID;
  static bool ok = allow_big_lds(attn_apply_kernel<1, 2>) && allow_big_lds(attn_apply_kernel<2, 1>) &&
                   allow_big_lds(attn_apply_kernel<2, 2>) && allow_big_lds(attn_apply_kernel<4, 1>) &&
                   allow_big_lds(attn_apply_kernel<4, 2>);
  (void)ok;
  dim3 g((p.Lq + T - 1) / T, p.B), blk(kThreads);
  hipStream_t st = pcr_s(stream);
  const bool wide = 2 * p.d > 128 || p.cout > 128 || p.cfinal > 128;   // some layer has > 4 cout blocks
  if (tb == 4) {
    if (wide) hipLaunchKernelGGL((attn_apply_kernel<4, 2>), g, blk, lds, st, a);
    else hipLaunchKernelGGL((attn_apply_kernel<4, 1>), g, blk, lds, st, a);
  } else if (tb == 2) {
    if (wide) hipLaunchKernelGGL((attn_apply_kernel<2, 2>), g, blk, lds, st, a);
    else hipLaunchKernelGGL((attn_apply_kernel<2, 1>), g, blk, lds, st, a);
  } else {
    hipLaunchKernelGGL((attn_apply_kernel<1, 2>), g, blk, lds, st, a);
  }
  PCR_CHECK_LAUNCH();
  return PCR_OK;
}

