// Filter gradient of the bf16 matrix-core CIN (see cin_bf16_kernels.hip for the family).  A file of
// its own: the kernel keeps 256 accumulator registers per lane, which this compiler only handles in
// its default (AGPR) MFMA form, so the Makefile builds this file without -amdgpu-mfma-vgpr-form.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "cin_bf16.h"
#include "xdfm_common.h"

using namespace xdfm_bf16;

namespace {

// ---------------------------------------------------------------- filter gradient
// dW[(f,h),n] = sum_m x0[f,m] xk[h,m] dcurr[n,m]: the contraction runs over m, so here m is the K of
// the matrix product.  A block = 4 waves = 4 consecutive f over one split of the m range; a wave
// keeps the whole [h][n] tile of its f in accumulators (HT x NT tiles of 32 x 32).  Per 64 m the
// block stages xk[h][m] and dcurr[n][m] as bf16 in LDS (shared by its waves, loaded one chunk ahead
// into registers; xk stays fp32 there); a wave forms its A operand x0[f,m] * xk[h,m] from the LDS
// fragment and its own f's factors (packed fp32 products, rounded to bf16 once), B = dcurr.  Partial tiles per split go to the
// workspace and are added in split order by cin_bf16_dw_reduce_kernel.
constexpr int kDwKM = 64;     // m per staged chunk (4 k steps)
constexpr int kDwPitch = 36;  // uint32 per LDS row of dcurr: 64 bf16 + 16 bytes, so 32 rows spread over all banks
constexpr int kDwXPitch = 68; // floats per LDS row of xk (kept fp32: the product with x0 needs no unpacking)

// FAST: D even, B*D a multiple of 64, H and L filling their tiles -- every staged element exists, so
// the staging is plain 8-byte loads at a per-chunk scalar base + one 32-bit offset per lane (the
// general form spends more instructions on masks and addresses than on everything else together),
// and the main loop fixes the order of what a wave issues (below).  Tiles not filled (rows past H or
// L, read again and stored as zeros) measured slower in this form than in the general one.
template <int HT, int NT, bool FAST>
__global__ __launch_bounds__(256, 1) void cin_bf16_dw_kernel(Dims c, const float* __restrict__ x0,
                                                              const float* __restrict__ xk,
                                                              const float* __restrict__ d_xnext,
                                                              float* __restrict__ part, int S, int OGn,
                                                              int chunks_per_split) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int kXs = HT * 32 * kDwXPitch, kDs = NT * 32 * kDwPitch, kBuf = kXs + kDs + 4 * kDwKM;  // 4-byte units
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
  // blocks of one split sit on one XCD (block id modulo 8) and share its L2 for the streamed inputs
  // (the grid holds 8 * ceil(S / 8) * OGn blocks; those whose split does not exist leave at once)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = xcd + 8 * (slot / OGn), og = slot % OGn;
  if (split >= S) return;
  const int f = og * 4 + wave;
  const int64_t total_chunks = (c.cols + kDwKM - 1) / kDwKM;
  const int64_t ch0 = (int64_t)split * chunks_per_split;
  int nch = (int)(total_chunks - ch0 < chunks_per_split ? total_chunks - ch0 : chunks_per_split);
  if (nch < 0) nch = 0;

  // staging: a thread owns one pair of m (columns 2*pc, 2*pc+1) of rows r0 + 8i
  const int pc = tid & 31, r0 = tid >> 5;
  float xv[HT * 4][2], dv[NT * 4][2], x0v;
  bool livef = false;
  // The blocks of a split stream the same bytes in step, so whoever asks first for a line takes the
  // HBM miss and the others wait on the same fill: every chunk costs a miss latency, more than its
  // products.  With D = 64 (a chunk is one sample: H and L rows of 256 B, each array contiguous)
  // each block therefore also touches its 1 / OGn share of the 64-byte pieces of the chunk two
  // ahead of the one being loaded; the value is never used (kept alive until the next chunk's
  // stores, when it has long arrived: loads return in order, so it costs no wait).
  const bool ahead = FAST && c.D == kDwKM;
  const int64_t ch_end = ch0 + nch;
  float pf_prev = 0.f, pf_cur = 0.f;
  // FAST: the loads of a chunk as single steps, so that the main loop can issue them one at a time
  // between matrix products (33 loads at once keep a wave in the issue queue of the CU's one
  // memory pipeline for about as long as half of its products take)
  const char *px = nullptr, *pd = nullptr;
  uint32_t vx = 0, vd = 0, step = 0;
  constexpr int kLoads = HT * 4 + NT * 4 + 1;
  auto fast_prep = [&](int64_t ch) __attribute__((always_inline)) {
    // the chunk's first sample gives a base every lane shares (a scalar pointer); what a lane adds
    // to it is a 32-bit byte offset (host checked), one add per load
    const uint32_t mbase = (uint32_t)(ch * kDwKM);  // host checked: B*D < 2^31
    const uint32_t bfirst = mbase / (uint32_t)c.D;
    const uint32_t m = mbase + 2 * pc;
    const uint32_t b0 = m / (uint32_t)c.D, d0 = m - b0 * (uint32_t)c.D;
    px = reinterpret_cast<const char*>(xk + (int64_t)bfirst * c.H * c.D);
    pd = reinterpret_cast<const char*>(d_xnext + (int64_t)bfirst * c.L * c.D);
    vx = ((b0 - bfirst) * (uint32_t)(c.H * c.D) + d0) * 4u;
    vd = ((b0 - bfirst) * (uint32_t)(c.L * c.D) + d0) * 4u;
    step = (uint32_t)c.D * 4u;  // bytes per row
  };
  auto fast_load = [&](int64_t ch, int i) __attribute__((always_inline)) {
    if (i < HT * 4) {
      const uint32_t row = (uint32_t)(r0 + 8 * i);
      const float2 v = *reinterpret_cast<const float2*>(px + (vx + row * step));
      xv[i][0] = v.x;
      xv[i][1] = v.y;
    } else if (i < HT * 4 + NT * 4) {
      const int k = i - HT * 4;
      const uint32_t row = (uint32_t)(r0 + 8 * k);
      const float2 v = *reinterpret_cast<const float2*>(pd + (vd + row * step));
      dv[k][0] = v.x;
      dv[k][1] = v.y;
    } else {
      const uint32_t mm = (uint32_t)(ch * kDwKM) + lane;
      const uint32_t bb = mm / (uint32_t)c.D;
      x0v = x0[((int64_t)bb * c.F + (f < c.F ? f : 0)) * c.D + (mm - bb * (uint32_t)c.D)];
      livef = f < c.F;
      if (ahead && ch + 2 < ch_end) {
        const uint32_t nx = (uint32_t)c.H * 4, nd = (uint32_t)c.L * 4;  // 64-byte pieces per sample
        uint32_t j = (uint32_t)og + (uint32_t)OGn * (uint32_t)tid;
        if (j >= nx + nd) j = (uint32_t)og;
        const float* p = j < nx ? xk + (ch + 2) * (int64_t)c.H * kDwKM + 16 * j
                                : d_xnext + (ch + 2) * (int64_t)c.L * kDwKM + 16 * (j - nx);
        pf_cur = *p;
      }
    }
  };
  auto issue = [&](int64_t ch) __attribute__((always_inline)) {
    if (FAST) {
      fast_prep(ch);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) fast_load(ch, i);
      return;
    }
    const int64_t m = ch * kDwKM + 2 * pc;
    const bool ok0 = m < c.cols, ok1 = m + 1 < c.cols;
    const int64_t b0 = ok0 ? m / c.D : 0;
    const int d0 = ok0 ? (int)(m - b0 * c.D) : 0;
    const int64_t b1 = (d0 + 1 < c.D) ? b0 : b0 + 1;
    const int d1 = (d0 + 1 < c.D) ? d0 + 1 : 0;
    if ((c.D & 1) == 0 && ok1) {  // an even m and its neighbour lie in one row: 8-byte loads
#pragma unroll
      for (int i = 0; i < HT * 4; ++i) {
        const int row = r0 + 8 * i;
        const float2 v = row < c.H ? *reinterpret_cast<const float2*>(xk + (b0 * c.H + row) * c.D + d0)
                                   : make_float2(0.f, 0.f);
        xv[i][0] = v.x;
        xv[i][1] = v.y;
      }
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        const int row = r0 + 8 * i;
        const float2 v = row < c.L ? *reinterpret_cast<const float2*>(d_xnext + (b0 * c.L + row) * c.D + d0)
                                   : make_float2(0.f, 0.f);
        dv[i][0] = v.x;
        dv[i][1] = v.y;
      }
    } else {
#pragma unroll
      for (int i = 0; i < HT * 4; ++i) {
        const int row = r0 + 8 * i;
        xv[i][0] = (ok0 && row < c.H) ? xk[(b0 * c.H + row) * c.D + d0] : 0.f;
        xv[i][1] = (ok1 && row < c.H) ? xk[(b1 * c.H + row) * c.D + d1] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        const int row = r0 + 8 * i;
        dv[i][0] = (ok0 && row < c.L) ? d_xnext[(b0 * c.L + row) * c.D + d0] : 0.f;
        dv[i][1] = (ok1 && row < c.L) ? d_xnext[(b1 * c.L + row) * c.D + d1] : 0.f;
      }
    }
    const int64_t mm = ch * kDwKM + lane;  // the block's four f: one factor per thread
    x0v = 0.f;
    if (mm < c.cols && f < c.F) {
      const int64_t b = mm / c.D;
      x0v = x0[(b * c.F + f) * c.D + (int)(mm - b * c.D)];
    }
  };
  // FAST: what fast_load(., i) brought, into the LDS image (every staged row exists)
  auto fast_store = [&](int buf, int i) __attribute__((always_inline)) {
    uint32_t* xs = lds + buf * kBuf;
    uint32_t* ds = xs + kXs;
    if (i < HT * 4) {
      *reinterpret_cast<float2*>(reinterpret_cast<float*>(xs) + (r0 + 8 * i) * kDwXPitch + 2 * pc) =
          make_float2(xv[i][0], xv[i][1]);
    } else if (i < HT * 4 + NT * 4) {
      const int k = i - HT * 4;
      ds[(r0 + 8 * k) * kDwPitch + pc] = pack_bf16(dv[k][0], dv[k][1]);
    } else {
      reinterpret_cast<float*>(ds + kDs)[wave * kDwKM + lane] = livef ? x0v : 0.f;
    }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
    uint32_t* xs = lds + buf * kBuf;
    uint32_t* ds = xs + kXs;
    float* xc = reinterpret_cast<float*>(ds + kDs);
    if (FAST) {
#pragma unroll
      for (int i = 0; i < kLoads; ++i) fast_store(buf, i);
      return;
    }
#pragma unroll
    for (int i = 0; i < HT * 4; ++i)
      *reinterpret_cast<float2*>(reinterpret_cast<float*>(xs) + (r0 + 8 * i) * kDwXPitch + 2 * pc) =
          make_float2(xv[i][0], xv[i][1]);
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) ds[(r0 + 8 * i) * kDwPitch + pc] = pack_bf16(dv[i][0], dv[i][1]);
    xc[wave * kDwKM + lane] = x0v;
  };

  f32x16 acc[HT][NT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ht][nt][i] = 0.f;

  // Chunk ch + 1 is loaded during the products of chunk ch and stored after them.  General form: all
  // loads first.  FAST: one load at a time between the products of the first two k steps (a burst of
  // 33 loads keeps the wave in the issue queue of the CU's one memory pipeline, matrix core idle,
  // for about as long as half of the products take; stores spread the same way, or loads spread
  // later than this, measured slower: the data is not there in time).
  if (nch > 0) {
    issue(ch0);
    stash(0);
  }
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const bool more = ch + 1 < nch;
    if (more) {
      if (FAST)
        fast_prep(ch0 + ch + 1);
      else
        issue(ch0 + ch + 1);
    }
    const uint32_t* xs = lds + (ch & 1) * kBuf;
    const uint32_t* ds = xs + kXs;
    const float* xc = reinterpret_cast<const float*>(ds + kDs) + wave * kDwKM;
    if (!FAST) {  // general form: the order is the compiler's (it spreads the loads' address work over the products)
#pragma unroll
      for (int ks = 0; ks < kDwKM / 16; ++ks) {
        typedef float f32x4g __attribute__((ext_vector_type(4)));
        const f32x4g m0 = *reinterpret_cast<const f32x4g*>(xc + 16 * ks + 8 * half);
        const f32x4g m1 = *reinterpret_cast<const f32x4g*>(xc + 16 * ks + 8 * half + 4);
        bf16x8 ag[HT];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
          const float* xr = reinterpret_cast<const float*>(xs) + (32 * ht + r) * kDwXPitch + 16 * ks + 8 * half;
          const f32x4g p0 = *reinterpret_cast<const f32x4g*>(xr) * m0;  // packed fp32 multiplies
          const f32x4g p1 = *reinterpret_cast<const f32x4g*>(xr + 4) * m1;
          uint4 z;
          z.x = pack_bf16(p0.x, p0.y);
          z.y = pack_bf16(p0.z, p0.w);
          z.z = pack_bf16(p1.x, p1.y);
          z.w = pack_bf16(p1.z, p1.w);
          ag[ht] = as_frag(z);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 bg = as_frag(*reinterpret_cast<const uint4*>(ds + (32 * nt + r) * kDwPitch + 8 * ks + 4 * half));
#pragma unroll
          for (int ht = 0; ht < HT; ++ht) acc[ht][nt] = mfma(ag[ht], bg, acc[ht][nt]);
        }
      }
    } else {
    // One wave per SIMD: nothing else hides an LDS read, so the operands of k step ks + 1 are read
    // before the matrix products of step ks start, and its A fragments are formed between the two
    // halves of those products (the matrix core works on while the VALU multiplies).
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    struct OperandSet {
      f32x4v m0, m1, x[HT][2];
      uint4 b[NT];
    };
    auto read_set = [&](int ks, OperandSet& o) __attribute__((always_inline)) {
      o.m0 = *reinterpret_cast<const f32x4v*>(xc + 16 * ks + 8 * half);
      o.m1 = *reinterpret_cast<const f32x4v*>(xc + 16 * ks + 8 * half + 4);
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const float* xr = reinterpret_cast<const float*>(xs) + (32 * ht + r) * kDwXPitch + 16 * ks + 8 * half;
        o.x[ht][0] = *reinterpret_cast<const f32x4v*>(xr);
        o.x[ht][1] = *reinterpret_cast<const f32x4v*>(xr + 4);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)  // 8 ks + 4 half: uint32 column of this lane's 8 m in the bf16 image
        o.b[nt] = *reinterpret_cast<const uint4*>(ds + (32 * nt + r) * kDwPitch + 8 * ks + 4 * half);
    };
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    // a quarter of one A fragment: one packed multiply, one packed conversion
    auto form_step = [&](const OperandSet& o, uint4 (&z)[HT], int j) {
      const int ht = j >> 2, q = j & 3;
      const f32x4v xv4 = o.x[ht][q >> 1], mv4 = (q >> 1) ? o.m1 : o.m0;
      const f32x2v p = (q & 1) ? __builtin_shufflevector(xv4, xv4, 2, 3) * __builtin_shufflevector(mv4, mv4, 2, 3)
                               : __builtin_shufflevector(xv4, xv4, 0, 1) * __builtin_shufflevector(mv4, mv4, 0, 1);
      const uint32_t w = pack_bf16(p.x, p.y);
      if (q == 0) z[ht].x = w;
      if (q == 1) z[ht].y = w;
      if (q == 2) z[ht].z = w;
      if (q == 3) z[ht].w = w;
    };
    bf16x8 a[HT], b[NT];
    {
      OperandSet o;
      uint4 z[HT];
      read_set(0, o);
#pragma unroll
      for (int j = 0; j < HT * 4; ++j) form_step(o, z, j);
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) a[ht] = as_frag(z[ht]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = as_frag(o.b[nt]);
    }
    // The order below is the order issued (the scheduler may not move anything across a barrier):
    // reads of step ks + 1, then each matrix product followed by its share of forming the next A
    // fragments, which the VALU does while the matrix core is busy with that product.
    constexpr int kStepsPerProduct = 4 / NT;  // HT * 4 quarter fragments over HT * NT products
#pragma unroll
    for (int ks = 0; ks < kDwKM / 16; ++ks) {
      const bool next = ks + 1 < kDwKM / 16;
      OperandSet o;
      uint4 z[HT];
      if (next) read_set(ks + 1, o);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
          acc[ht][nt] = mfma(a[ht], b[nt], acc[ht][nt]);
          __builtin_amdgcn_sched_barrier(0);
          if (FAST && more) {
            constexpr int kHalf = 2 * HT * NT;  // products of two k steps
            const int g = (ks * NT + nt) * HT + ht;
            if (g < kHalf) {
#pragma unroll
              for (int i = g * kLoads / kHalf; i < (g + 1) * kLoads / kHalf; ++i) fast_load(ch0 + ch + 1, i);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (next) {
#pragma unroll
            for (int k = 0; k < kStepsPerProduct; ++k) form_step(o, z, (nt * HT + ht) * kStepsPerProduct + k);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (next) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) a[ht] = as_frag(z[ht]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = as_frag(o.b[nt]);
      }
    }
    }
    if (more) stash((ch + 1) & 1);
    asm volatile("" ::"v"(pf_prev));
    pf_prev = pf_cur;
    __syncthreads();
  }
  if (f < c.F) {
    float* dst = part + ((size_t)split * (OGn * 4) + f) * (HT * 32) * (NT * 32);
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          dst[(size_t)(32 * ht + tile_row(i, half)) * (NT * 32) + 32 * nt + r] = acc[ht][nt][i];
  }
}

// the partial sums of one element, added in split order (the same bits on every run), eight loads in flight
__device__ __forceinline__ float sum_splits(const float* __restrict__ src, size_t stride, int S) {
  float sum = 0.f;
  int k = 0;
  for (; k + 8 <= S; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; k < S; ++k) sum += src[(size_t)k * stride];
  return sum;
}

__global__ void cin_bf16_dw_reduce_kernel(Dims c, const float* __restrict__ part, int S, int OGn, int Hp, int Lp,
                                          float* __restrict__ d_w, int ldw) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)c.F * c.H * c.L) return;
  const int n = (int)(idx % c.L);
  const int64_t t = idx / c.L;
  const int h = (int)(t % c.H), f = (int)(t / c.H);
  if (c.masked && h <= f) {  // not part of the layer: its gradient is zero
    d_w[(int64_t)(f * c.H + h) * ldw + n] = 0.f;
    return;
  }
  const size_t stride = (size_t)(OGn * 4) * Hp * Lp;
  const float* src = part + ((size_t)f * Hp + h) * Lp + n;
  const float sum = sum_splits(src, stride, S);
  d_w[(int64_t)(f * c.H + h) * ldw + n] = c.masked ? 2.f * sum : sum;
}

// splits of the m range for the filter gradient: at least four staged chunks each
void dw_geometry(int64_t cols, int F, int* S, int* OGn, int* chunks_per_split) {
  const int64_t chunks = (cols + kDwKM - 1) / kDwKM;
  *OGn = (F + 3) / 4;
  // one block per CU at a time (256 accumulators per lane) and the blocks of a split share an XCD:
  // per XCD (32 CUs) at most 64 blocks, i.e. two full rounds, in whole splits
  int64_t s = 8 * (64 / *OGn > 0 ? 64 / *OGn : 1);
  if (s > chunks / 4) s = chunks / 4;
  if (s < 1) s = 1;
  *S = (int)s;
  *chunks_per_split = (int)((chunks + s - 1) / s);
}


// ---------------------------------------------------------------- full tiles, D = 64: loads that write LDS themselves
// With D = 64 a chunk is one sample: xk[b] and dcurr[b] are 128 rows of 256 contiguous bytes each.  They go to
// LDS as they are (fp32, `global_load_lds_dwordx4`: no staging registers, no store pass with the matrix core
// idle): a wave-instruction fills 1 KB of LDS in lane order, so the image is the linear sequence of 16-byte
// pieces of rows of 17 pieces (272 bytes = kDwXPitch words: the 17th is padding and gets the 16th's bytes again),
// 34 instructions per array, dealt to the four waves.  dcurr is rounded to bf16 when a fragment is read
// (packed conversions between the products, as the A operand's multiplies are).  The x0 factors of a wave's f
// take one 4-byte instruction per wave.  The barrier at a chunk's end waits for the loads of the next.
constexpr int kGlPieces = 17;                                   // 16-byte pieces per LDS row
constexpr int kGlInstr = 128 * kGlPieces / 64;                  // wave-instructions per array and chunk: 34
constexpr int kGlPerWave = (kGlInstr + 3) / 4;                  // 9 (waves 2 and 3: 8)
constexpr int kGlImg = 128 * kDwXPitch;                         // words per array image
constexpr int kGlBuf = 2 * kGlImg + 4 * kDwKM;                  // words per buffer: xk, dcurr, factors
#ifndef GL_EVERY
#define GL_EVERY 2
#endif
constexpr int kGlEvery = GL_EVERY;

typedef __attribute__((address_space(1))) const void* gl_src_t;
typedef __attribute__((address_space(3))) void* gl_dst_t;

__global__ __launch_bounds__(256, 1) void cin_bf16_dw_glds_kernel(Dims c, const float* __restrict__ x0,
                                                                   const float* __restrict__ xk,
                                                                   const float* __restrict__ d_xnext,
                                                                   float* __restrict__ part, int S, int OGn,
                                                                   int chunks_per_split) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int HT = 4, NT = 4;
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = xcd + 8 * (slot / OGn), og = slot % OGn;
  if (split >= S) return;
  const int f = og * 4 + wave;
  const int64_t total_chunks = c.B;  // one sample each
  const int64_t ch0 = (int64_t)split * chunks_per_split;
  int nch = (int)(total_chunks - ch0 < chunks_per_split ? total_chunks - ch0 : chunks_per_split);
  if (nch < 0) nch = 0;

  // byte offset, inside a sample's array, of the piece this lane brings with its wave's i-th instruction
  uint32_t soff[kGlPerWave];
#pragma unroll
  for (int i = 0; i < kGlPerWave; ++i) {
    const int q = (wave + 4 * i) * 64 + lane;
    const int row = q / kGlPieces, col = q - row * kGlPieces;
    soff[i] = (uint32_t)(row * 256 + (col < 16 ? col : 15) * 16);
  }
  const int64_t x0_row = (int64_t)(f < c.F ? f : c.F - 1) * kDwKM + lane;  // this wave's factors: x0[b][f][lane]
  // instruction i of this wave for sample b into buffer buf: i < 9 xk, i < 18 dcurr, 18 the factors
  constexpr int kIssues = 2 * kGlPerWave + 1;
  auto issue = [&](int64_t b, int buf, int i) __attribute__((always_inline)) {
    unsigned char* base = smem + (size_t)buf * kGlBuf * 4;
    if (i < 2 * kGlPerWave) {
      const int arr = i / kGlPerWave, k = i - arr * kGlPerWave;
      const int j = wave + 4 * k;
      if (j < kGlInstr) {
        const char* src = reinterpret_cast<const char*>((arr ? d_xnext : xk) + b * 128 * kDwKM) + soff[k];
        __builtin_amdgcn_global_load_lds((gl_src_t)src, (gl_dst_t)(base + (size_t)arr * kGlImg * 4 + (size_t)j * 1024),
                                         16, 0, 0);
      }
    } else {
      __builtin_amdgcn_global_load_lds((gl_src_t)(x0 + b * c.F * kDwKM + x0_row),
                                       (gl_dst_t)(base + (size_t)2 * kGlImg * 4 + (size_t)wave * kDwKM * 4), 4, 0, 0);
    }
  };

  f32x16 acc[HT][NT];
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ht][nt][i] = 0.f;

  if (nch > 0) {
#pragma unroll
    for (int i = 0; i < kIssues; ++i) issue(ch0, 0, i);
  }
  __syncthreads();  // (with loads to LDS outstanding the compiler waits for them here)
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  struct OperandSet {
    f32x4v m0, m1, x[HT][2], d[NT][2];
  };
  for (int ch = 0; ch < nch; ++ch) {
    const bool more = ch + 1 < nch;
    const float* xs = reinterpret_cast<const float*>(lds + (ch & 1) * kGlBuf);
    const float* ds = xs + kGlImg;
    const float* xc = ds + kGlImg + wave * kDwKM;
    auto read_set = [&](int ks, OperandSet& o) __attribute__((always_inline)) {
      o.m0 = *reinterpret_cast<const f32x4v*>(xc + 16 * ks + 8 * half);
      o.m1 = *reinterpret_cast<const f32x4v*>(xc + 16 * ks + 8 * half + 4);
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        const float* xr = xs + (32 * ht + r) * kDwXPitch + 16 * ks + 8 * half;
        o.x[ht][0] = *reinterpret_cast<const f32x4v*>(xr);
        o.x[ht][1] = *reinterpret_cast<const f32x4v*>(xr + 4);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float* dr = ds + (32 * nt + r) * kDwXPitch + 16 * ks + 8 * half;
        o.d[nt][0] = *reinterpret_cast<const f32x4v*>(dr);
        o.d[nt][1] = *reinterpret_cast<const f32x4v*>(dr + 4);
      }
    };
    // quarter fragments: j < 16 the A operand (x0 factor times xk, then rounded), j >= 16 the B operand (rounded)
    auto form_step = [&](const OperandSet& o, uint4* za, uint4* zb, int j) __attribute__((always_inline)) {
      const int t = (j & 15) >> 2, q = j & 3;
      uint32_t w;
      if (j < 16) {
        const f32x4v xv4 = o.x[t][q >> 1], mv4 = (q >> 1) ? o.m1 : o.m0;
        const f32x2v pr = (q & 1) ? __builtin_shufflevector(xv4, xv4, 2, 3) * __builtin_shufflevector(mv4, mv4, 2, 3)
                                  : __builtin_shufflevector(xv4, xv4, 0, 1) * __builtin_shufflevector(mv4, mv4, 0, 1);
        w = pack_bf16(pr.x, pr.y);
      } else {
        const f32x4v dv4 = o.d[t][q >> 1];
        w = (q & 1) ? pack_bf16(dv4.z, dv4.w) : pack_bf16(dv4.x, dv4.y);
      }
      uint4* z = j < 16 ? za : zb;
      if (q == 0) z[t].x = w;
      if (q == 1) z[t].y = w;
      if (q == 2) z[t].z = w;
      if (q == 3) z[t].w = w;
    };
    bf16x8 a[HT], b[NT];
    {
      OperandSet o;
      uint4 za[HT], zb[NT];
      read_set(0, o);
#pragma unroll
      for (int j = 0; j < 32; ++j) form_step(o, za, zb, j);
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) a[ht] = as_frag(za[ht]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = as_frag(zb[nt]);
    }
#pragma unroll
    for (int ks = 0; ks < kDwKM / 16; ++ks) {
      const bool next = ks + 1 < kDwKM / 16;
      OperandSet o;
      uint4 za[HT], zb[NT];
      if (next) read_set(ks + 1, o);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
          acc[ht][nt] = mfma(a[ht], b[nt], acc[ht][nt]);
          __builtin_amdgcn_sched_barrier(0);
          const int g = (ks * NT + nt) * HT + ht;  // the next sample's loads: one per kGlEvery products
          if (more && g % kGlEvery == 0 && g / kGlEvery < kIssues) {
            issue(ch0 + ch + 1, (ch + 1) & 1, g / kGlEvery);
            __builtin_amdgcn_sched_barrier(0);
          }

          if (next) {
            form_step(o, za, zb, 2 * (nt * HT + ht));
            form_step(o, za, zb, 2 * (nt * HT + ht) + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (next) {
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) a[ht] = as_frag(za[ht]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = as_frag(zb[nt]);
      }
    }
    __syncthreads();  // everyone is done reading this buffer; the next one's loads have landed
  }
  if (f < c.F) {
    float* dst = part + ((size_t)split * (OGn * 4) + f) * (HT * 32) * (NT * 32);
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          dst[(size_t)(32 * ht + tile_row(i, half)) * (NT * 32) + 32 * nt + r] = acc[ht][nt][i];
  }
}

// ---------------------------------------------------------------- filter gradient of the masked first layer
// xk == x0 and only the pairs f < h are filter rows (xDeepFM.py:208-217), 741 of the 39 x 64 (f, h) slots the
// kernel above computes for configs[4].  Here the PAIRS are the rows of the product: pair p = (f, h) in row-major
// order of the triangle, 32 pairs per tile, kTriTT tiles per wave (a lane's A row is its pair: x0[f,m] * x0[h,m]
// from two LDS rows), B = dcurr as above.  Two blocks instead of ten read each staged chunk, a third of the
// matrix products remain.  Same loop as the FAST form above: operands a k step ahead, loads between products.
constexpr int kTriTT = 3;  // 32-pair tiles per wave: 192 accumulators with NT = 4
constexpr int kTriXR = 64; // staged x0 rows (host: F <= 64)

__host__ __device__ inline int tri_pairs(int F) { return F * (F - 1) / 2; }

template <int NT>
__global__ __launch_bounds__(256, 1) void cin_bf16_dw_tri_kernel(Dims c, const float* __restrict__ x0,
                                                                  const float* __restrict__ d_xnext,
                                                                  float* __restrict__ part, int S, int OGT,
                                                                  int chunks_per_split) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int TT = kTriTT;
  constexpr int kXs = kTriXR * kDwXPitch, kDs = NT * 32 * kDwPitch, kBuf = kXs + kDs;  // 4-byte units
  uint32_t* lds = reinterpret_cast<uint32_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, r = lane & 31;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int split = xcd + 8 * (slot / OGT), og = slot % OGT;
  if (split >= S) return;
  const int P = tri_pairs(c.F), Ppad = OGT * 4 * TT * 32;
  const int64_t total_chunks = c.cols / kDwKM;  // host checked: a multiple of 64
  const int64_t ch0 = (int64_t)split * chunks_per_split;
  int nch = (int)(total_chunks - ch0 < chunks_per_split ? total_chunks - ch0 : chunks_per_split);
  if (nch < 0) nch = 0;

  // this lane's pairs: LDS word offsets of their two rows (a pair past the end repeats the last one; its
  // results are not stored)
  int offf[TT], offh[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    int p = ((og * 4 + wave) * TT + t) * 32 + r;
    if (p >= P) p = P - 1;
    int f = 0, rem = p;
    while (rem >= c.F - 1 - f) {
      rem -= c.F - 1 - f;
      ++f;
    }
    offf[t] = f * kDwXPitch;
    offh[t] = (f + 1 + rem) * kDwXPitch;
  }

  // staging as in the FAST form: a thread owns columns 2 pc, 2 pc + 1 of rows r0 + 8 i
  const int pc = tid & 31, r0 = tid >> 5;
  constexpr int XL = kTriXR / 8, kLoads = XL + NT * 4;
  float xv[XL][2], dv[NT * 4][2];
  const char *px = nullptr, *pd = nullptr;
  uint32_t vx = 0, vd = 0, step = 0;
  auto prep = [&](int64_t ch) __attribute__((always_inline)) {
    const uint32_t mbase = (uint32_t)(ch * kDwKM);  // host checked: B*D < 2^31
    const uint32_t bfirst = mbase / (uint32_t)c.D;
    const uint32_t m = mbase + 2 * pc;
    const uint32_t b0 = m / (uint32_t)c.D, d0 = m - b0 * (uint32_t)c.D;
    px = reinterpret_cast<const char*>(x0 + (int64_t)bfirst * c.F * c.D);
    pd = reinterpret_cast<const char*>(d_xnext + (int64_t)bfirst * c.L * c.D);
    vx = ((b0 - bfirst) * (uint32_t)(c.F * c.D) + d0) * 4u;
    vd = ((b0 - bfirst) * (uint32_t)(c.L * c.D) + d0) * 4u;
    step = (uint32_t)c.D * 4u;
  };
  auto load = [&](int i) __attribute__((always_inline)) {  // rows past the end: the last row once more, zeroed when stored
    if (i < XL) {
      const uint32_t row = min((uint32_t)(r0 + 8 * i), (uint32_t)(c.F - 1));
      const float2 v = *reinterpret_cast<const float2*>(px + (vx + row * step));
      xv[i][0] = v.x;
      xv[i][1] = v.y;
    } else {
      const int k = i - XL;
      const uint32_t row = min((uint32_t)(r0 + 8 * k), (uint32_t)(c.L - 1));
      const float2 v = *reinterpret_cast<const float2*>(pd + (vd + row * step));
      dv[k][0] = v.x;
      dv[k][1] = v.y;
    }
  };
  auto store = [&](int buf, int i) __attribute__((always_inline)) {
    uint32_t* xs = lds + buf * kBuf;
    uint32_t* ds = xs + kXs;
    if (i < XL) {
      const bool in = r0 + 8 * i < c.F;
      *reinterpret_cast<float2*>(reinterpret_cast<float*>(xs) + (r0 + 8 * i) * kDwXPitch + 2 * pc) =
          make_float2(in ? xv[i][0] : 0.f, in ? xv[i][1] : 0.f);
    } else {
      const int k = i - XL;
      const bool in = r0 + 8 * k < c.L;
      ds[(r0 + 8 * k) * kDwPitch + pc] = pack_bf16(in ? dv[k][0] : 0.f, in ? dv[k][1] : 0.f);
    }
  };

  f32x16 acc[TT][NT];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][nt][i] = 0.f;

  if (nch > 0) {
    prep(ch0);
#pragma unroll
    for (int i = 0; i < kLoads; ++i) load(i);
#pragma unroll
    for (int i = 0; i < kLoads; ++i) store(0, i);
  }
  __syncthreads();
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  struct OperandSet {
    f32x4v xf[TT][2], xh[TT][2];
    uint4 b[NT];
  };
  for (int ch = 0; ch < nch; ++ch) {
    const bool more = ch + 1 < nch;
    if (more) prep(ch0 + ch + 1);
    const float* xs = reinterpret_cast<const float*>(lds + (ch & 1) * kBuf);
    const uint32_t* ds = lds + (ch & 1) * kBuf + kXs;
    auto read_set = [&](int ks, OperandSet& o) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float* pf = xs + offf[t] + 16 * ks + 8 * half;
        const float* ph = xs + offh[t] + 16 * ks + 8 * half;
        o.xf[t][0] = *reinterpret_cast<const f32x4v*>(pf);
        o.xf[t][1] = *reinterpret_cast<const f32x4v*>(pf + 4);
        o.xh[t][0] = *reinterpret_cast<const f32x4v*>(ph);
        o.xh[t][1] = *reinterpret_cast<const f32x4v*>(ph + 4);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        o.b[nt] = *reinterpret_cast<const uint4*>(ds + (32 * nt + r) * kDwPitch + 8 * ks + 4 * half);
    };
    auto form_step = [&](const OperandSet& o, uint4* z, int j) __attribute__((always_inline)) {
      const int t = j >> 2, q = j & 3;
      const f32x4v a4 = o.xf[t][q >> 1], h4 = o.xh[t][q >> 1];
      const f32x2v pr = (q & 1) ? __builtin_shufflevector(a4, a4, 2, 3) * __builtin_shufflevector(h4, h4, 2, 3)
                                : __builtin_shufflevector(a4, a4, 0, 1) * __builtin_shufflevector(h4, h4, 0, 1);
      const uint32_t w = pack_bf16(pr.x, pr.y);
      if (q == 0) z[t].x = w;
      if (q == 1) z[t].y = w;
      if (q == 2) z[t].z = w;
      if (q == 3) z[t].w = w;
    };
    bf16x8 a[TT], b[NT];
    {
      OperandSet o;
      uint4 z[TT];
      read_set(0, o);
#pragma unroll
      for (int j = 0; j < TT * 4; ++j) form_step(o, z, j);
#pragma unroll
      for (int t = 0; t < TT; ++t) a[t] = as_frag(z[t]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = as_frag(o.b[nt]);
    }
    constexpr int kStepsPerProduct = 4 / NT;  // TT * 4 quarter fragments over TT * NT products
#pragma unroll
    for (int ks = 0; ks < kDwKM / 16; ++ks) {
      const bool next = ks + 1 < kDwKM / 16;
      OperandSet o;
      uint4 z[TT];
      if (next) read_set(ks + 1, o);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          acc[t][nt] = mfma(a[t], b[nt], acc[t][nt]);
          __builtin_amdgcn_sched_barrier(0);
          if (more) {  // the next chunk's loads, one at a time between the products of the first two k steps
            constexpr int kHalf = 2 * TT * NT;
            const int g = (ks * NT + nt) * TT + t;
            if (g < kHalf) {
#pragma unroll
              for (int i = g * kLoads / kHalf; i < (g + 1) * kLoads / kHalf; ++i) load(i);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (next) {
#pragma unroll
            for (int k = 0; k < kStepsPerProduct; ++k) form_step(o, z, (nt * TT + t) * kStepsPerProduct + k);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (next) {
#pragma unroll
        for (int t = 0; t < TT; ++t) a[t] = as_frag(z[t]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = as_frag(o.b[nt]);
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < kLoads; ++i) store((ch + 1) & 1, i);
    }
    __syncthreads();
  }
  // part[split][pair (Ppad)][NT * 32]
  float* dst = part + (size_t)split * Ppad * (NT * 32);
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int pbase = ((og * 4 + wave) * TT + t) * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int p = pbase + tile_row(i, half);
        if (p < P) dst[(size_t)p * (NT * 32) + 32 * nt + r] = acc[t][nt][i];
      }
  }
}

__global__ void cin_bf16_dw_tri_reduce_kernel(Dims c, const float* __restrict__ part, int S, int Ppad, int Lp,
                                              float* __restrict__ d_w, int ldw) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)c.F * c.H * c.L) return;
  const int n = (int)(idx % c.L);
  const int64_t t = idx / c.L;
  const int h = (int)(t % c.H), f = (int)(t / c.H);
  if (h <= f) {  // not part of the layer: its gradient is zero
    d_w[(int64_t)(f * c.H + h) * ldw + n] = 0.f;
    return;
  }
  const int p = f * (2 * c.F - f - 1) / 2 + (h - f - 1);
  const float* src = part + (size_t)p * Lp + n;
  const size_t stride = (size_t)Ppad * Lp;
  const float sum = sum_splits(src, stride, S);
  d_w[(int64_t)(f * c.H + h) * ldw + n] = 2.f * sum;
}

// the pair form serves: the masked layer, F <= 64 staged rows, every staged m existing
bool tri_serves(int64_t cols, int F, int D, int L, int masked) {
  static const bool on = !(getenv("XDFM_DW_TRI") && getenv("XDFM_DW_TRI")[0] == '0');
  return on && masked && F >= 2 && F <= kTriXR && (D & 1) == 0 && cols % kDwKM == 0 && cols < ((int64_t)1 << 31) &&
         (int64_t)(F > L ? F : L) * D * 4 * (kDwKM / D + 3) < ((int64_t)1 << 31);
}

void tri_geometry(int64_t cols, int F, int* S, int* OGT, int* chunks_per_split) {
  const int64_t chunks = cols / kDwKM;
  const int tiles = (tri_pairs(F) + 31) / 32;
  *OGT = (tiles + 4 * kTriTT - 1) / (4 * kTriTT);
  int64_t s = 8 * (32 / *OGT > 0 ? 32 / *OGT : 1);  // one block per CU, the blocks of a split on one XCD
  if (s > chunks / 4) s = chunks / 4;
  if (s < 1) s = 1;
  *S = (int)s;
  *chunks_per_split = (int)((chunks + s - 1) / s);
}

}  // namespace

extern "C" int xdfm_cin_bf16_bwd_dw_workspace_floats(int64_t B, int32_t F, int32_t H, int32_t D, int32_t L,
                                                     size_t* floats) {
  XDFM_REQUIRE(floats && B >= 0 && F > 0 && H > 0 && D > 0 && L > 0, "xdfm_cin_bf16_bwd_dw_workspace_floats: bad argument");
  XDFM_REQUIRE(H <= 128 && L <= 128, "xdfm_cin_bf16: at most 128 inputs per side and 128 channels per call (H=%d, L=%d)", H, L);
  int S, OGn, cps;
  dw_geometry(B * D, F, &S, &OGn, &cps);
  const int HT = H <= 32 ? 1 : (H <= 64 ? 2 : 4), NT = L <= 32 ? 1 : (L <= 64 ? 2 : 4);
  *floats = (size_t)S * OGn * 4 * HT * 32 * NT * 32;
  if (F == H && F >= 2 && F <= kTriXR) {  // the masked layer's pair form (whether it serves is decided per call)
    int St, OGT, cpst;
    tri_geometry(B * D / kDwKM * kDwKM, F, &St, &OGT, &cpst);
    const size_t tri = (size_t)St * OGT * 4 * kTriTT * 32 * NT * 32;
    if (tri > *floats) *floats = tri;
  }
  return XDFM_OK;
}

extern "C" int xdfm_cin_bf16_bwd_dw(const float* x0, const float* xk, const float* d_xnext, int64_t B, int32_t F,
                                    int32_t H, int32_t D, int32_t L, int32_t masked, float* d_w, int32_t ldw,
                                    float* workspace, void* stream) {
  XDFM_REQUIRE(H <= 128 && L <= 128, "xdfm_cin_bf16: at most 128 inputs per side and 128 channels per call (H=%d, L=%d)", H, L);
  Dims c;
  int rc = make_dims(B, F, H, D, L, masked, 8, &c);
  if (rc) return rc;
  XDFM_REQUIRE(d_w, "xdfm_cin_bf16_bwd_dw: null d_w");
  if (ldw == 0) ldw = L;
  int S, OGn, cps;
  dw_geometry(c.cols, F, &S, &OGn, &cps);
  const int HT = H <= 32 ? 1 : (H <= 64 ? 2 : 4), NT = L <= 32 ? 1 : (L <= 64 ? 2 : 4);
  XDFM_REQUIRE(B == 0 || (x0 && xk && d_xnext && workspace), "xdfm_cin_bf16_bwd_dw: null argument");
  if (B > 0 && tri_serves(c.cols, F, D, L, masked)) {
    int St, OGT, cpst;
    tri_geometry(c.cols, F, &St, &OGT, &cpst);
    const size_t ldst = (size_t)2 * (kTriXR * kDwXPitch + NT * 32 * kDwPitch) * 4;
    const dim3 gridt((unsigned)(8 * ((St + 7) / 8) * OGT));
#define TRI(NTV)                                                                                                   \
  do {                                                                                                             \
    if (ldst > 64 * 1024)                                                                                          \
      XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bf16_dw_tri_kernel<NTV>),                     \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));                        \
    XDFM_LAUNCH("cin_bf16_dw_tri_kernel<" #NTV ">", stream, (cin_bf16_dw_tri_kernel<NTV>), gridt, dim3(256), ldst, \
                c, x0, d_xnext, workspace, St, OGT, cpst);                                                         \
  } while (0)
    if (NT == 1)
      TRI(1);
    else if (NT == 2)
      TRI(2);
    else
      TRI(4);
#undef TRI
    const int64_t total = (int64_t)F * H * L;
    XDFM_LAUNCH("cin_bf16_dw_tri_reduce_kernel", stream, cin_bf16_dw_tri_reduce_kernel,
                dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c, workspace, St, OGT * 4 * kTriTT * 32, NT * 32,
                d_w, ldw);
    XDFM_LAUNCH_CHECK();
    return XDFM_OK;
  }
  if (B == 0) S = 0;
  static const bool glds_on = !(getenv("XDFM_DW_GLDS") && getenv("XDFM_DW_GLDS")[0] == '0');
  if (glds_on && S > 0 && !masked && D == kDwKM && H == 128 && L == 128 &&
      (reinterpret_cast<uintptr_t>(xk) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_xnext) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(x0) & 3) == 0) {
    const size_t ldsg = (size_t)2 * kGlBuf * 4;
    const dim3 gridg((unsigned)(8 * ((S + 7) / 8) * OGn));
    XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bf16_dw_glds_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg));
    XDFM_LAUNCH("cin_bf16_dw_glds_kernel", stream, cin_bf16_dw_glds_kernel, gridg, dim3(256), ldsg, c, x0, xk, d_xnext,
                workspace, S, OGn, cps);
    const int64_t total = (int64_t)F * H * L;
    XDFM_LAUNCH("cin_bf16_dw_reduce_kernel", stream, cin_bf16_dw_reduce_kernel, dim3((unsigned)((total + 255) / 256)),
                dim3(256), 0, c, workspace, S, OGn, HT * 32, NT * 32, d_w, ldw);
    XDFM_LAUNCH_CHECK();
    return XDFM_OK;
  }
  const size_t lds = (size_t)2 * (HT * 32 * kDwXPitch + NT * 32 * kDwPitch + 4 * kDwKM) * 4;
  const dim3 grid((unsigned)(8 * ((S + 7) / 8) * OGn));
  const bool fast = (D & 1) == 0 && H == HT * 32 && L == NT * 32 && (c.cols % kDwKM) == 0 &&
                    c.cols < ((int64_t)1 << 31) &&
                    // a lane's byte offset from its chunk's first sample: at most 64 / D + 1 samples on
                    (int64_t)(H > L ? H : L) * D * 4 * (kDwKM / D + 3) < ((int64_t)1 << 31);
#define DW_M(HTV, NTV, FASTV)                                                                                        \
  do {                                                                                                               \
    if (lds > 64 * 1024)                                                                                             \
      XDFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bf16_dw_kernel<HTV, NTV, FASTV>),               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
    XDFM_LAUNCH("cin_bf16_dw_kernel<" #HTV ", " #NTV ">", stream, (cin_bf16_dw_kernel<HTV, NTV, FASTV>), grid,       \
                dim3(256), lds, c, x0, xk, d_xnext, workspace, S, OGn, cps);                                         \
  } while (0)
#define DW(HTV, NTV)                 \
  do {                               \
    if (fast)                        \
      DW_M(HTV, NTV, true);          \
    else                             \
      DW_M(HTV, NTV, false);         \
  } while (0)
#define DW_N(HTV)            \
  do {                       \
    if (NT == 1)             \
      DW(HTV, 1);            \
    else if (NT == 2)        \
      DW(HTV, 2);            \
    else                     \
      DW(HTV, 4);            \
  } while (0)
  if (S > 0) {
    if (HT == 1)
      DW_N(1);
    else if (HT == 2)
      DW_N(2);
    else
      DW_N(4);
  }
#undef DW_N
#undef DW
#undef DW_M
  const int64_t total = (int64_t)F * H * L;
  XDFM_LAUNCH("cin_bf16_dw_reduce_kernel", stream, cin_bf16_dw_reduce_kernel, dim3((unsigned)((total + 255) / 256)),
              dim3(256), 0, c, workspace, S, OGn, HT * 32, NT * 32, d_w, ldw);
  XDFM_LAUNCH_CHECK();
  return XDFM_OK;
}
