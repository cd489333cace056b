// gemm16 v2 — same contract as gemm16.hip (sfm_gemm16), restructured for latency hiding:
//   * operands go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), no VGPR staging;
//     the buffer descriptor's range check supplies the conv zero padding and the M tail
//     (out-of-range rows/positions read as 0), so the loop has no predicates;
//   * BK = 64 k-tiles in a STAGES-deep LDS ring, counted `s_waitcnt vmcnt(N)` + ONE raw
//     `s_barrier` per k-tile: STAGES-1 tiles stay in flight under the MFMAs;
//   * LDS rows are 128 B; the 16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7)
//     (applied on the SOURCE address, since LDS-DMA writes lane-linear), which makes the
//     ds_read_b128 fragment reads conflict-free;
//   * epilogue: accumulators -> per-wave fp32 LDS image -> 8-column vectors per thread:
//     bias / activation / GLU / residual / GroupNorm partials, then 16-byte row stores;
//   * XCD-aware block order: the N-tiles of one M-tile (same A rows) run on one XCD (L2 reuse).
#include "sfm_common.h"
#include <cstdlib>

#include "gemm16_epi.h"

// BM = 128: wave tile 64 x WN, 2 workgroups/CU.  BM = 256: wave tile 128 x WN - every B fragment read from LDS feeds
// 4 MFMAs instead of 2 (0.75 instead of 1 ds_read_b128 per MFMA: the 128-row kernel needs the full 128 B/clk of the LDS
// at the matrix cores' rate), at 1 workgroup/CU.
template <class T, int BN, int STAGES, int BM>
__global__ __launch_bounds__(256) void gemm16v2_kernel(Gemm2Params p) {
  constexpr int BKB = 128;                           // k-tile: 64 elements = 128 bytes per row
  constexpr int WN = BN / 2;                         // wave tile columns
  constexpr int NJ = WN / 32;
  constexpr int MI = BM / 64;                        // 32-row blocks per wave
  constexpr int A_STAGE = BM * BKB, B_STAGE = BN * BKB, STAGE = A_STAGE + B_STAGE;
  constexpr int NA = BM / 32, NB = BN / 32;          // LDS-DMA instructions per wave per tile
  constexpr int NLD = NA + NB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hl = lane >> 5;

  // ---- XCD-aware tile order ----
  const int total = gridDim.x;
  int id = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ntile = id % p.nNt;
  const int rest = id / p.nNt;
  const int mtile = rest % p.nMt;
  const int b = rest / p.nMt;
  const int n0 = ntile * BN, l0 = mtile * BM;

  auto a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)b * p.a_batch_stride), 0, p.a_records, 0x00020000);
  auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_records, 0x00020000);

  // ---- per-lane source coordinates of the LDS-DMA pieces (8 rows x 128 B per instruction) ----
  int a_rowoff[NA], a_swz[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave * NA + i) * 8 + (lane >> 3);
    const int pos0 = (l0 + row) * p.stride - p.pad;           // may be negative: wraps out of range -> 0
    a_rowoff[i] = pos0 * p.lda * 2;                           // byte offset of the row's first element
    a_swz[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;           // logical k element held by this lane's chunk
  }
  int b_rowoff[NB], b_swz[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave * NB + i) * 8 + (lane >> 3);
    b_rowoff[i] = (n0 + row) * p.Kpad * 2;
    b_swz[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  const bool contiguous = (p.lda == p.Cin) || (p.cin_shift >= 30);

  auto issue = [&](int kt, int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = kt * 64 + a_swz[i];
      int eoff = contiguous ? j : ((j >> p.cin_shift) * p.lda + (j & (p.Cin - 1)));
      // positions outside [0, Lin) of this batch fall outside the descriptor range (negative offsets
      // wrap to > 2^31) and read as zero: that IS the conv zero padding; k beyond K is forced out of range
      int voff = a_rowoff[i] + eoff * 2;
      if (j >= p.K) voff = -1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(sa + (wave * NA + i) * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int voff = b_rowoff[i] + (kt * 64 + b_swz[i]) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(sb + (wave * NB + i) * 1024), 16, voff, 0, 0, 0);
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nt = p.Kpad >> 6;
  constexpr int D = STAGES - 1;                        // prefetch distance
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nt) issue(s, s);

  // fragment read offsets (bytes) inside a stage
  int fa_off[MI][4], fb_off[NJ][4];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * (BM / 2) + i * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_off[i][s] = row * BKB + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * WN + j * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fb_off[j][s] = A_STAGE + row * BKB + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }

  int stage = 0;
  for (int t = 0; t < nt; ++t) {
    // tile t must have landed; up to D-1 younger tiles may stay in flight
    const int younger = (nt - 1 - t) < (D - 1) ? (nt - 1 - t) : (D - 1);
    if (younger >= 2) wait_ring<2 * NLD>();              // (+ lgkmcnt(0): the reads of the stage refilled below have returned)
    else if (younger == 1) wait_ring<NLD>();
    else wait_ring<0>();
    __builtin_amdgcn_s_barrier();
    if (t + D < nt) {
      int st = stage + D;
      if (st >= STAGES) st -= STAGES;
      issue(t + D, st);
    }
    const unsigned char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[MI], fb[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sbase + fa_off[i][s]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sbase + fb_off[j][s]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma(fb[j], fa[i], acc[i][j]);   // W as the A operand: the TRANSPOSED tile
    }
    if (++stage == STAGES) stage = 0;
  }

  // ------------------------------ epilogue ------------------------------
  __syncthreads();                                     // every wave is done reading the ring
  constexpr int IMG_LD = WN + 4;                       // floats per image row
  float* img = reinterpret_cast<float*>(smem) + wave * (64 * IMG_LD);
#pragma unroll
  for (int hp = 0; hp < MI / 2; ++hp) {                // 64 rows of the wave tile per pass
  if (hp > 0) {
    __builtin_amdgcn_s_waitcnt(0xC07F);                // the previous pass has finished reading the image
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)                        // transposed accumulators: a register quad = 4 consecutive columns of
        *reinterpret_cast<f32x4*>(&img[(i * 32 + l31) * IMG_LD + j * 32 + 8 * q + 4 * hl]) =   // output row l31: 16 ds_write_b128
            f32x4{acc[2 * hp + i][j][4 * q], acc[2 * hp + i][j][4 * q + 1], acc[2 * hp + i][j][4 * q + 2],     // per tile instead
                  acc[2 * hp + i][j][4 * q + 3]};                                                              // of 64 ds_write_b32
  __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the wave's own image is complete
  __builtin_amdgcn_wave_barrier();
  const int row_base = l0 + wm * (BM / 2) + hp * 64;   // first output row of this pass

  const bool glu = (p.epi == EPI_GLU);
  const int ecols = glu ? 32 : WN;                     // image columns that produce outputs
  const int cpr = ecols >> 3;                          // 8-column chunks per row
  const int rpp = 64 / cpr;                            // rows per pass
  const int c8 = (lane % cpr) * 8, rsub = lane / cpr;
  const int colb = n0 + wn * WN;                       // first packed column of this wave
  const int ncol0 = glu ? ((colb >> 1) + c8) : (colb + c8);   // first output column of this thread
  float bia[8], big[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bia[e] = p.bias ? p.bias[colb + c8 + e] : 0.f;     // bias is padded to Npad
    big[e] = (glu && p.bias) ? p.bias[colb + 32 + c8 + e] : 0.f;
  }
  const long long obase = (long long)b * p.o_batch_stride;
  // Residual epilogue (out = resid + alpha * drop(acc + bias), no statistics: attention out_proj, pointwise2, the second FFN
  // Linear of the unfused / training path): all of the tile's residual rows are requested BEFORE the first one is used.  In the
  // general loop below each pass loads its 32 bytes and waits for them (a global round trip per 8 rows: the epilogue of these
  // GEMMs took 23 us of their 37 at M 51 264).
  // The same lean loop serves the plain epilogue (EPI_NONE: bias, one rounding, 16-byte stores - the input-gradient GEMMs of the
  // training step, Q | K | V): the general loop below spends ~560 VALU instructions per wave tile on its per-element predicates
  // and mode switches (profiles/r03/pmc_gemm16v2.json: 10.9 VALU per MFMA at K = 256), this one about a third of that.
  // With GroupNorm partials (the conv forward of the training step) the whole wave tile has to be inside N: the sums are folded
  // across the wave's lanes below, every lane takes part.
  if ((p.epi == EPI_RESID || p.epi == EPI_NONE) && !glu && p.vec_ok &&
      (p.gn_partial ? (colb + WN <= p.N) : (ncol0 + 8 <= p.N))) {
    constexpr int CPR = WN >> 3, RPP = 64 / CPR;
    const bool has_res = (p.epi == EPI_RESID);
    float gs = 0.f, gq = 0.f;                          // GroupNorm partials of this lane's 8 columns over its rows
    const int c8s = (lane % CPR) * 8, rs = lane / CPR;
    const int nc = colb + c8s;
    f32x4 rr[CPR][2];
    if (has_res) {
#pragma unroll
      for (int it = 0; it < CPR; ++it) {
        int m = row_base + it * RPP + rs;
        m = m < p.Lout ? m : p.Lout - 1;               // clamped: loaded, not used
        const float* rp = p.resid + (long long)b * p.r_batch_stride + (long long)m * p.ldr + nc;
        rr[it][0] = *reinterpret_cast<const f32x4*>(rp);
        rr[it][1] = *reinterpret_cast<const f32x4*>(rp + 4);
      }
    }
    f32x4 bb0 = {0.f, 0.f, 0.f, 0.f}, bb1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      bb0 = *reinterpret_cast<const f32x4*>(p.bias + nc);
      bb1 = *reinterpret_cast<const f32x4*>(p.bias + nc + 4);
    }
#pragma unroll
    for (int it = 0; it < CPR; ++it) {
      const int row = it * RPP + rs;
      const int m = row_base + row;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + c8s]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + c8s + 4]);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = x0[e] + bb0[e];
        y[4 + e] = x1[e] + bb1[e];
      }
      if (p.gn_partial && m < p.Lout) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { gs += y[e]; gq = __builtin_fmaf(y[e], y[e], gq); }   // (fmaf: profiles/README.md, "the lost sums of squares")
      }
      if (has_res) {
        if (p.p_drop > 0.f) {                            // residual-branch dropout, same counters as sfm_ew_train mode 4
          float kp[8];
          sfm_keep_scale8(p.seed, ((unsigned long long)b * p.Lout + (m < p.Lout ? m : 0)) * p.N + nc, p.p_drop, 1.0f / (1.0f - p.p_drop), kp);
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] *= kp[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = rr[it][0][e] + p.alpha * y[e];
          y[4 + e] = rr[it][1][e] + p.alpha * y[4 + e];
        }
      }
      if (m >= p.Lout) continue;
      const long long orow = obase + (long long)m * p.ldo + nc;
      if (p.out_f32 == 1) {
        float* op = reinterpret_cast<float*>(p.out) + orow;
        *reinterpret_cast<f32x4*>(op) = f32x4{y[0], y[1], y[2], y[3]};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{y[4], y[5], y[6], y[7]};
      } else {
        u32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack2_out<T>(y[2 * e], y[2 * e + 1], p.out_f32 == 2);
        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.out) + orow) = pk;
      }
    }
    if (p.gn_partial) {
      // lanes with the same column chunk hold different rows: fold them, then fold the chunks of a group
      for (int o = CPR; o < 64; o <<= 1) {
        gs += __shfl_xor(gs, o, 64);
        gq += __shfl_xor(gq, o, 64);
      }
      const int cpg = p.gn_group >> 3;                 // 8-column chunks per group (1, 2 or 4)
      for (int o = 1; o < cpg; o <<= 1) {
        gs += __shfl_xor(gs, o, 64);
        gq += __shfl_xor(gq, o, 64);
      }
      if (lane < CPR && (lane % cpg) == 0 && (row_base >> 6) < p.gn_slots) {
        const int ngroups = p.N / p.gn_group;           // one partial per 64-row block of the output, whatever BM is
        const long long slot = ((long long)b * p.gn_slots + (row_base >> 6)) * ngroups + nc / p.gn_group;
        p.gn_partial[slot * 2 + 0] = gs;
        p.gn_partial[slot * 2 + 1] = gq;
      }
    }
    continue;                                          // next 64-row pass of the wave tile
  }
  // Fused Swish of the FFN in training (EPI_SWISH_DUAL / EPI_SWISH_BWD: gemm16_epi.h).  Backward: ALL of the tile's saved
  // derivative factors are requested before the first is used (in the strip epilogue of the wide / persistent kernels each of the
  // 8 passes waits out its own global round trip: 0.53 ms per launch at M 205 056, N 1024 against 0.19 ms of HBM time).
  // Forward: sigmoid by v_exp + v_rcp (1 ulp) instead of the division sequence; u and d from one sigmoid.
  if ((p.epi == EPI_SWISH_BWD || p.epi == EPI_SWISH_DUAL) && ncol0 + 8 <= p.N) {
    constexpr int CPR = WN >> 3, RPP = 64 / CPR;
    const int c8s = (lane % CPR) * 8, rs = lane / CPR;
    const int nc = colb + c8s;
    u32x4 ax[CPR];
    if (p.epi == EPI_SWISH_BWD) {
#pragma unroll
      for (int it = 0; it < CPR; ++it) {
        int m = row_base + it * RPP + rs;
        m = m < p.Lout ? m : p.Lout - 1;               // clamped: loaded, not used
        ax[it] = *reinterpret_cast<const u32x4*>(p.aux + obase + (long long)m * p.ldo + nc);
      }
    }
    f32x4 bb0 = {0.f, 0.f, 0.f, 0.f}, bb1 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      bb0 = *reinterpret_cast<const f32x4*>(p.bias + nc);
      bb1 = *reinterpret_cast<const f32x4*>(p.bias + nc + 4);
    }
    const float ik = (p.p_drop > 0.f) ? 1.0f / (1.0f - p.p_drop) : 1.0f;
#pragma unroll
    for (int it = 0; it < CPR; ++it) {
      const int row = it * RPP + rs;
      const int m = row_base + row;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + c8s]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + c8s + 4]);
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = x0[e] + bb0[e];
        y[4 + e] = x1[e] + bb1[e];
      }
      const long long orow = obase + (long long)m * p.ldo + nc;
      if (p.epi == EPI_SWISH_BWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[2 * e] *= T::to_f32((u16)(ax[it][e] & 0xffffu));
          y[2 * e + 1] *= T::to_f32((u16)(ax[it][e] >> 16));
        }
      } else {
        float kp[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (p.p_drop > 0.f)
          sfm_keep_scale8(p.seed, ((unsigned long long)b * p.Lout + (m < p.Lout ? m : 0)) * p.N + nc, p.p_drop, ik, kp);
        float dd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-y[e]));
          const float ks = kp[e] * sg;
          dd[e] = ks * (1.0f + y[e] * (1.0f - sg));
          y[e] = y[e] * ks;
        }
        if (m < p.Lout) {
          u32x4 pd;
#pragma unroll
          for (int e = 0; e < 4; ++e) pd[e] = pack2<T>(dd[2 * e], dd[2 * e + 1]);
          *reinterpret_cast<u32x4*>(p.out2 + orow) = pd;
        }
      }
      if (m >= p.Lout) continue;
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = pack2<T>(y[2 * e], y[2 * e + 1]);
      *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.out) + orow) = pk;
    }
    continue;                                          // next 64-row pass of the wave tile
  }
  float gsum = 0.f, gsq = 0.f;
  for (int r0 = 0; r0 < 64; r0 += rpp) {
    const int row = r0 + rsub;
    const int m = row_base + row;
    float v[8];
    {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + c8]);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + c8 + 4]);
      v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
      v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bia[e];
    if (glu) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + 32 + c8]);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(&img[row * IMG_LD + 32 + c8 + 4]);
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= __builtin_amdgcn_rcpf(1.0f + __expf(-(g[e] + big[e])));   // v_rcp (1 ulp), not the division sequence
    }
    const bool mok = m < p.Lout;
    if (p.gn_partial && mok) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ncol0 + e < p.N) { gsum += v[e]; gsq = __builtin_fmaf(v[e], v[e], gsq); }     // (fmaf: see profiles/README.md, "the lost sums of squares")
    }
    switch (p.epi) {
      case EPI_SWISH:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = swish_f(v[e]);
        break;
      case EPI_GELU:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        break;
      case EPI_SIGMOID:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sigmoid_f(v[e]);
        break;
      case EPI_TANH_SCALE:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.alpha * tanhf(v[e]);
        break;
      case EPI_SIGMA:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = expf(0.5f * fminf(fmaxf(v[e], -10.f), 10.f));
        break;
      case EPI_CPEA:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (ncol0 + e < p.nsplit) ? sigmoid_f(v[e]) : p.alpha * tanhf(v[e]);
        break;
      default: break;
    }
    if (!mok) continue;
    const long long orow = obase + (long long)m * p.ldo + ncol0;
    if (p.vec_ok && ncol0 + 8 <= p.N) {
      if (p.epi == EPI_RESID) {
        if (p.p_drop > 0.f) {                           // residual-branch dropout, same counters as sfm_ew_train mode 4
          const float ik = 1.0f / (1.0f - p.p_drop);
          const unsigned long long e0 = ((unsigned long long)b * p.Lout + m) * p.N + ncol0;
float kp[8];
          sfm_keep_scale8(p.seed, e0, p.p_drop, ik, kp);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= kp[e];
        }
        const float* rp = p.resid + (long long)b * p.r_batch_stride + (long long)m * p.ldr + ncol0;
        const f32x4 r0v = *reinterpret_cast<const f32x4*>(rp);
        const f32x4 r1v = *reinterpret_cast<const f32x4*>(rp + 4);
        v[0] = r0v[0] + p.alpha * v[0]; v[1] = r0v[1] + p.alpha * v[1];
        v[2] = r0v[2] + p.alpha * v[2]; v[3] = r0v[3] + p.alpha * v[3];
        v[4] = r1v[0] + p.alpha * v[4]; v[5] = r1v[1] + p.alpha * v[5];
        v[6] = r1v[2] + p.alpha * v[6]; v[7] = r1v[3] + p.alpha * v[7];
      }
      if (p.out_f32 == 1) {
        float* op = reinterpret_cast<float*>(p.out) + orow;
        f32x4 a = {v[0], v[1], v[2], v[3]}, c = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(op) = a;
        *reinterpret_cast<f32x4*>(op + 4) = c;
      } else {
        u32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack2_out<T>(v[2 * e], v[2 * e + 1], p.out_f32 == 2);
        *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.out) + orow) = pk;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (ncol0 + e < p.N) {
          float y = v[e];
          if (p.epi == EPI_RESID) {
            if (p.p_drop > 0.f)
              y *= sfm_keep_scale(p.seed, ((unsigned long long)b * p.Lout + m) * p.N + ncol0 + e, p.p_drop, 1.0f / (1.0f - p.p_drop));
            y = p.resid[(long long)b * p.r_batch_stride + (long long)m * p.ldr + ncol0 + e] + p.alpha * y;
          }
          if (p.out_f32 == 1) reinterpret_cast<float*>(p.out)[orow + e] = y;
          else reinterpret_cast<u16*>(p.out)[orow + e] = from_f32_out<T>(y, p.out_f32 == 2);
        }
      }
    }
  }
  if (p.gn_partial) {
    // lanes with the same column chunk hold different rows: fold them, then fold the chunks of a group
    for (int o = cpr; o < 64; o <<= 1) {
      gsum += __shfl_xor(gsum, o, 64);
      gsq += __shfl_xor(gsq, o, 64);
    }
    const int cpg = p.gn_group >> 3;                   // 8-column chunks per group (1, 2 or 4)
    for (int o = 1; o < cpg; o <<= 1) {
      gsum += __shfl_xor(gsum, o, 64);
      gsq += __shfl_xor(gsq, o, 64);
    }
    if (lane < cpr && (lane % cpg) == 0 && ncol0 < p.N && (row_base >> 6) < p.gn_slots) {
      const int ngroups = p.N / p.gn_group;             // one partial per 64-row block of the output, whatever BM is
      const long long slot = ((long long)b * p.gn_slots + (row_base >> 6)) * ngroups + ncol0 / p.gn_group;
      p.gn_partial[slot * 2 + 0] = gsum;
      p.gn_partial[slot * 2 + 1] = gsq;
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form (variant 6): 128 x BN tiles, 2-stage ring, 2 workgroups per CU that each walk a list of tiles.
// The skinny GEMMs of the path (K = 64 .. 1280, i.e. 1 .. 20 k-tiles per output tile) spend as long in workgroup launch,
// the first loads' latency and the epilogue as in the MFMA loop; here the first k-tile of the NEXT tile is already in
// flight while a tile's epilogue runs (the epilogue image is a separate, per-wave 8-row LDS strip, so the ring stays
// free), and there is one workgroup launch per CU slot instead of one per tile.
// Tile order: XCD x owns the contiguous range [x Q, (x+1) Q) of the (batch, M-tile, N-tile) list and its resident
// workgroups take neighbouring tiles, so the N-tiles of one M-tile share that XCD's L2.
template <class T, int BN>
__global__ __launch_bounds__(256) void gemm16p_kernel(Gemm2Params p, int total_tiles) {
  constexpr int BM = 128, BKB = 128;
  constexpr int WN = BN / 2, NJ = WN / 32;
  constexpr int A_STAGE = BM * BKB, B_STAGE = BN * BKB, STAGE = A_STAGE + B_STAGE;
  constexpr int NA = BM / 32, NB = BN / 32;
  constexpr int IMG_LD = WN + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hl = lane >> 5;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int Q = (total_tiles + 7) >> 3;
  const int t_begin = xcd * Q;
  const int t_end = min(total_tiles, t_begin + Q);
  if (t_begin + slot >= t_end) return;                 // whole workgroup leaves together: no barrier is ever missed

  auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_records, 0x00020000);
  const bool contiguous = (p.lda == p.Cin) || (p.cin_shift >= 30);
  const int nt = p.Kpad >> 6;

  // ---- issue side: (tile, k-tile) of the next LDS-DMA batch and its per-lane source coordinates ----
  int tile_i = t_begin + slot, ki = 0;
  int a_rowoff[NA], a_swz[NA], b_rowoff[NB], b_swz[NB];
  const u16* a_base = p.A;
  auto setup_issue = [&](int id) {
    const int ntile = id % p.nNt, rest = id / p.nNt;
    const int mtile = rest % p.nMt, bb = rest / p.nMt;
    a_base = p.A + (long long)bb * p.a_batch_stride;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (wave * NA + i) * 8 + (lane >> 3);
      const int pos0 = (mtile * BM + row) * p.stride - p.pad;
      a_rowoff[i] = pos0 * p.lda * 2;
      a_swz[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (wave * NB + i) * 8 + (lane >> 3);
      b_rowoff[i] = (ntile * BN + row) * p.Kpad * 2;
      b_swz[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    }
  };
  auto issue_next = [&](int stage) {
    auto a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, p.a_records, 0x00020000);
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = ki * 64 + a_swz[i];
      const int eoff = contiguous ? j : ((j >> p.cin_shift) * p.lda + (j & (p.Cin - 1)));
      int voff = a_rowoff[i] + eoff * 2;
      if (j >= p.K) voff = -1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(sa + (wave * NA + i) * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int voff = b_rowoff[i] + (ki * 64 + b_swz[i]) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(sb + (wave * NB + i) * 1024), 16, voff, 0, 0, 0);
    }
    if (++ki == nt) {
      ki = 0;
      tile_i += nslots;
      if (tile_i < t_end) setup_issue(tile_i);
    }
  };

  // fragment read offsets (bytes) inside a stage
  int fa_off[2][4], fb_off[NJ][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_off[i][s] = row * BKB + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * WN + j * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fb_off[j][s] = A_STAGE + row * BKB + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }
  float* img = reinterpret_cast<float*>(smem + 2 * STAGE) + wave * (8 * IMG_LD);

  setup_issue(tile_i);
  issue_next(0);
  int stage = 0;
  for (int tile = t_begin + slot; tile < t_end; tile += nslots) {
    const int ntile = tile % p.nNt, rest = tile / p.nNt;
    const int mtile = rest % p.nMt, b = rest / p.nMt;
    const int n0 = ntile * BN, l0 = mtile * BM;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < nt; ++t) {
      wait_ring<0>();                                  // this wave's share of the current k-tile has landed, its reads of the other stage returned
      __builtin_amdgcn_s_barrier();                    // ... everyone's have, and the other stage is no longer being read
      if (tile_i < t_end) issue_next(stage ^ 1);
      const unsigned char* sbase = smem + stage * STAGE;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u32x4 fa[2], fb[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sbase + fa_off[i][s]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sbase + fb_off[j][s]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
      }
      stage ^= 1;
    }

    gemm16_epilogue_strips<T, NJ>(p, acc, img, lane, b, n0 + wn * WN, l0 + wm * 64);
  }
  wait_vmcnt<0>();                                     // nothing of the ring is in flight when the workgroup retires
}

template <class T, int BN>
static int launch_p(const Gemm2Params& p, hipStream_t stream) {
  constexpr int lds = 2 * (128 + BN) * 128 + 4 * 8 * (BN / 2 + 4) * 4;
  static bool attr_set_dev[64] = {false};        // hipFuncSetAttribute is per device
  int attr_dev_ = 0;
  if (hipGetDevice(&attr_dev_) != hipSuccess || attr_dev_ < 0 || attr_dev_ >= 64) return SFM_ERR_LAUNCH;
  bool& attr_set = attr_set_dev[attr_dev_];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm16p_kernel<T, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return SFM_ERR_LAUNCH;
    attr_set = true;
  }
  const int total = p.nMt * p.nNt * p.B;
  int nblk = 512;                                      // 2 workgroups on each of the 256 CUs
  if (total < nblk) nblk = (total + 7) / 8 * 8;
  SFM_LAUNCH((gemm16p_kernel<T, BN>), dim3(nblk), dim3(256), lds, stream, p, total);
  return SFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Wide form (variant 9): 256 x BN tiles computed by NW = 8 (BN 128) or 16 (BN 256) waves, each wave a 64 x 64 tile as in the
// default kernel, 2-stage ring, one workgroup per CU.  The path's GEMMs stream their operands L2 -> LDS at 8-13 TB/s whatever
// the schedule (tools/gemm_probe.py); a 128 x 128 tile needs 15.6 B per kFLOP of that stream, 256 x 128 needs 11.7 and
// 256 x 256 needs 7.8.
template <class T, int BM, int BN, int NW>
__global__ __launch_bounds__(NW * 64) void gemm16w_kernel(Gemm2Params p) {
  constexpr int BKB = 128;
  constexpr int WGN = NW / (BM / 64);                  // waves along N (BM / 64 along M)
  constexpr int WN = BN / WGN, NJ = WN / 32;
  static_assert(WN == 64, "wave tile is 64 x 64");
  constexpr int A_STAGE = BM * BKB, B_STAGE = BN * BKB, STAGE = A_STAGE + B_STAGE;
  constexpr int NA = BM / 8 / NW, NB = BN / 8 / NW;    // LDS-DMA instructions (8 rows x 128 B) per wave per k-tile
  constexpr int IMG_LD = WN + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, hl = lane >> 5;

  const int total = gridDim.x;
  int id = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = id & 7, slot = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int ntile = id % p.nNt;
  const int rest = id / p.nNt;
  const int mtile = rest % p.nMt;
  const int b = rest / p.nMt;
  const int n0 = ntile * BN, l0 = mtile * BM;

  auto a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)b * p.a_batch_stride), 0, p.a_records, 0x00020000);
  auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.w_records, 0x00020000);

  int a_rowoff[NA], a_swz[NA], b_rowoff[NB], b_swz[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave * NA + i) * 8 + (lane >> 3);
    const int pos0 = (l0 + row) * p.stride - p.pad;
    a_rowoff[i] = pos0 * p.lda * 2;
    a_swz[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (wave * NB + i) * 8 + (lane >> 3);
    b_rowoff[i] = (n0 + row) * p.Kpad * 2;
    b_swz[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  const bool contiguous = (p.lda == p.Cin) || (p.cin_shift >= 30);

  auto issue = [&](int kt, int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = kt * 64 + a_swz[i];
      const int eoff = contiguous ? j : ((j >> p.cin_shift) * p.lda + (j & (p.Cin - 1)));
      int voff = a_rowoff[i] + eoff * 2;
      if (j >= p.K) voff = -1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(sa + (wave * NA + i) * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int voff = b_rowoff[i] + (kt * 64 + b_swz[i]) * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(sb + (wave * NB + i) * 1024), 16, voff, 0, 0, 0);
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int fa_off[2][4], fb_off[NJ][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fa_off[i][s] = row * BKB + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wn * WN + j * 32 + l31;
#pragma unroll
    for (int s = 0; s < 4; ++s) fb_off[j][s] = A_STAGE + row * BKB + (((2 * s + hl) ^ ((row >> 1) & 7)) << 4);
  }

  const int nt = p.Kpad >> 6;
  issue(0, 0);
  int stage = 0;
  for (int t = 0; t < nt; ++t) {
    wait_ring<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 1 < nt) issue(t + 1, stage ^ 1);
    const unsigned char* sbase = smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 fa[2], fb[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const u32x4*>(sbase + fa_off[i][s]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const u32x4*>(sbase + fb_off[j][s]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
    }
    stage ^= 1;
  }
  __syncthreads();                                     // every wave is done reading the ring: it becomes the epilogue strips
  float* img = reinterpret_cast<float*>(smem) + wave * (8 * IMG_LD);
  gemm16_epilogue_strips<T, NJ>(p, acc, img, lane, b, n0 + wn * WN, l0 + wm * 64);
}

template <class T, int BM, int BN, int NW>
static int launch_w(const Gemm2Params& p, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * 128;
  static bool attr_set_dev[64] = {false};        // hipFuncSetAttribute is per device
  int attr_dev_ = 0;
  if (hipGetDevice(&attr_dev_) != hipSuccess || attr_dev_ < 0 || attr_dev_ >= 64) return SFM_ERR_LAUNCH;
  bool& attr_set = attr_set_dev[attr_dev_];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm16w_kernel<T, BM, BN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return SFM_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid(p.nMt * p.nNt * p.B), block(NW * 64);
  SFM_LAUNCH((gemm16w_kernel<T, BM, BN, NW>), grid, block, lds, stream, p);
  return SFM_OK;
}

template <class T, int BN, int STAGES, int BM>
static int launch_v2(const Gemm2Params& p, hipStream_t stream) {
  constexpr int ring = STAGES * (BM + BN) * 128, image = 4 * 64 * (BN / 2 + 4) * 4;
  constexpr int lds = ring > image ? ring : image;
  static bool attr_set_dev[64] = {false};        // hipFuncSetAttribute is per device
  int attr_dev_ = 0;
  if (hipGetDevice(&attr_dev_) != hipSuccess || attr_dev_ < 0 || attr_dev_ >= 64) return SFM_ERR_LAUNCH;
  bool& attr_set = attr_set_dev[attr_dev_];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm16v2_kernel<T, BN, STAGES, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return SFM_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid(p.nMt * p.nNt * p.B), block(256);
  SFM_LAUNCH((gemm16v2_kernel<T, BN, STAGES, BM>), grid, block, lds, stream, p);
  return SFM_OK;
}

// contract: include/sincformer_hip.h (sfm_gemm16); `variant`: 0 = auto, 2 = 128 x 128(64) tiles, LDS-DMA ring with 2 stages
// (2 workgroups/CU), 6 = persistent form of it (tile loop inside the workgroup, next tile prefetched under the epilogue),
// 9 = 256 x 256 / 256 x 128 tiles on 16 / 8 waves, 10 = 512 x 128 tiles on 16 waves.  (The register-staged v1 kernel, the
// 3-stage ring and the 256-row wave tiles of round 1 were measured slower on every shape of the path and are gone:
// profiles/README.md.)  Operands of 2 GiB or more (buffer descriptors are 32-bit) are cut into row chunks here.
// sfm_gemm16_train = sfm_gemm16_ex + residual-branch dropout in the EPI_RESID epilogue (training forward):
// out = resid + alpha * keep(seed, m*N + n) / (1 - p_drop) * (A W^T + bias)
static int gemm16_impl(const void* A, const void* W, const float* bias, void* out, const float* resid,
                       float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                       long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                       int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                       int nsplit, int dtype, int variant, float p_drop, unsigned int seed, const void* aux, void* out2,
                       void* stream) {
  if (!A || !W || !out) return SFM_ERR_ARG;
  const bool swish = (epi == EPI_SWISH_DUAL || epi == EPI_SWISH_BWD);
  if (p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && epi != EPI_RESID && !swish)) return SFM_ERR_SHAPE;
  if (swish) {                                         // strip-epilogue kernels only; 16-bit, 16-byte aligned row vectors
    if ((epi == EPI_SWISH_DUAL && !out2) || (epi == EPI_SWISH_BWD && !aux)) return SFM_ERR_ARG;
    if (out_f32 != 0 || (N % 8) != 0 || (ldo % 8) != 0 || (o_batch_stride % 8) != 0 || gn_partial) return SFM_ERR_SHAPE;
    if ((((uintptr_t)out) % 16) != 0 || (out2 && (((uintptr_t)out2) % 16) != 0) || (aux && (((uintptr_t)aux) % 16) != 0))
      return SFM_ERR_SHAPE;
    if (variant != 2 && variant != 6 && variant != 9 && variant != 10) variant = 0;
  }
  if (B <= 0 || Lout <= 0 || N <= 0 || out_f32 < 0 || out_f32 > 2) return SFM_ERR_SHAPE;
  const long long a_rec = ((long long)(Lin - 1) * lda + Cin) * 2;
  const long long w_rec = (long long)Npad * Kpad * 2;
  if ((Kpad % 64) != 0 || (Npad % 64) != 0 || w_rec >= (1LL << 31) || (epi == EPI_GLU && Npad % 128 != 0) ||
      (gn_partial && gn_group != 8 && gn_group != 16 && gn_group != 32))
    return SFM_ERR_SHAPE;
  if (a_rec >= (1LL << 31) || (long long)Lout * stride * lda * 2 >= (1LL << 31)) {
    // the A operand does not fit one 32-bit buffer descriptor: plain GEMMs (one batch entry, 1 tap) are cut into row chunks
    // (no dropout on this path: the backward rebuilds the keep mask from the seed and the GLOBAL (row, column) index, a
    //  per-chunk re-seed with row indices restarting at 0 would give it another mask)
    if (B != 1 || ksize != 1 || stride != 1 || pad != 0 || gn_partial || swish || p_drop > 0.f) return SFM_ERR_SHAPE;
    const long long rows_max = ((1LL << 30) / ((long long)lda * 2)) & ~255LL;
    const int osz_ = out_f32 == 1 ? 4 : 2;
    for (long long r0 = 0; r0 < Lout; r0 += rows_max) {
      const int rows = (int)((Lout - r0 < rows_max) ? (Lout - r0) : rows_max);
      const int rc = gemm16_impl((const u16*)A + r0 * lda, W, bias, (char*)out + r0 * ldo * osz_, resid ? resid + r0 * ldr : nullptr,
                                 nullptr, 1, rows, rows, Cin, lda, 1, 1, 0, 0, Kpad, N, Npad, ldo, 0, ldr, 0, alpha, epi, out_f32,
                                 gn_group, nsplit, dtype, variant, 0.f, seed, nullptr, nullptr, stream);
      if (rc != SFM_OK) return rc;
    }
    return SFM_OK;
  }
  if (Cin % 8 != 0 || lda % 8 != 0 || lda < Cin) return SFM_ERR_SHAPE;
  const int K = ksize * Cin;
  if (K > Kpad) return SFM_ERR_SHAPE;
  int shift = 30;
  if (ksize > 1) {
    if (Cin & (Cin - 1)) return SFM_ERR_SHAPE;
    shift = 0;
    while ((1 << shift) < Cin) ++shift;
  }
  if (epi == EPI_RESID && !resid) return SFM_ERR_ARG;
  if (gn_partial && (N % gn_group) != 0) return SFM_ERR_SHAPE;
  if (epi == EPI_GLU && Npad != 2 * N) return SFM_ERR_SHAPE;
  Gemm2Params p;
  p.A = (const u16*)A; p.W = (const u16*)W; p.bias = bias; p.out = out; p.resid = resid; p.gn_partial = gn_partial;
  p.a_batch_stride = a_batch_stride; p.o_batch_stride = o_batch_stride; p.r_batch_stride = r_batch_stride;
  p.B = B; p.Lout = Lout; p.Lin = Lin; p.Cin = Cin; p.lda = lda; p.stride = stride; p.pad = pad; p.cin_shift = shift;
  p.K = K; p.Kpad = Kpad; p.N = N; p.Npad = Npad; p.ldo = ldo; p.ldr = ldr; p.alpha = alpha; p.epi = epi;
  p.out_f32 = out_f32; p.gn_group = gn_group; p.nsplit = nsplit; p.p_drop = p_drop; p.seed = seed;
  p.aux = (const u16*)aux; p.out2 = (u16*)out2;
  p.a_records = (int)a_rec; p.w_records = (int)w_rec;
  const int osz = out_f32 == 1 ? 4 : 2;
  const bool o_al = (((uintptr_t)out) % 16 == 0) && ((ldo * osz) % 16 == 0) && ((o_batch_stride * osz) % 16 == 0);
  const bool r_al = (epi != EPI_RESID) || ((((uintptr_t)resid) % 16 == 0) && ((ldr * 4) % 16 == 0) && ((r_batch_stride * 4) % 16 == 0));
  p.vec_ok = (o_al && r_al) ? 1 : 0;
  const bool bn128 = (Npad % 128 == 0);
  const int BNv = bn128 ? 128 : 64;
  // 256-row tiles on 8 or 16 waves (variant 9) and 512 x 128 tiles on 16 waves (variant 10) stay selectable, but auto no longer
  // picks them for the inference GEMMs: since the PerceptionAgent convs left for conv16p the shapes that remain are K = 256 /
  // 1024 linears, and there the 128 x 128 kernel wins on every one (tools/gemm_bench.py, round 2: M 51264, K 256, N 768:
  // 49 us against 75 us wide, 108 us wide inside the pass; bench c2 8.00 -> 7.83 ms per step).  The fused-Swish epilogues of the
  // training step still run best on the wide tiles (52.3 ms per step against 55.4 ms on the persistent 128 x 128 kernel).
  const long long tiles256 = (long long)((Lout + 255) / 256) * (Npad / 256) * B;
  const bool auto_wide = (variant == 0) && swish && (Npad % 256 == 0) && tiles256 >= 512;
  const bool auto_tall = false;
  const bool tall = ((variant == 10) && bn128) || auto_tall;
  const bool wide = ((variant == 9) && bn128) || auto_wide;
  const bool wide256 = wide && (Npad % 256 == 0);
  const int BMv = tall ? 512 : (wide ? 256 : 128);
  p.nMt = (Lout + BMv - 1) / BMv;
  p.nNt = wide256 ? Npad / 256 : Npad / BNv;
  p.gn_slots = 2 * ((Lout + 127) / 128);               // partial slots per batch entry: one per 64 output rows (padded to 128)
  hipStream_t st = (hipStream_t)stream;
  // the persistent kernel is 5-20 % faster than variant 2 on isolated launches of the path's skinny GEMMs
  // (tools/gemm_bench.py) but 2 % slower inside the forward pass (bench.py, same box, A/B/A/B): not the default
  const bool persistent = (variant == 6) || (swish && !wide && !tall && variant != 2);
#define GO(TT)                                                                                            \
  if (tall) return launch_w<TT, 512, 128, 16>(p, st);                                                   \
  if (wide) return wide256 ? launch_w<TT, 256, 256, 16>(p, st) : launch_w<TT, 256, 128, 8>(p, st);      \
  if (persistent) return bn128 ? launch_p<TT, 128>(p, st) : launch_p<TT, 64>(p, st);                    \
  if (bn128) return launch_v2<TT, 128, 2, 128>(p, st);                                                   \
  else return launch_v2<TT, 64, 2, 128>(p, st);
  if (dtype == SFM_DT_BF16) { GO(BF16) }
  if (dtype == SFM_DT_F16) { GO(F16) }
#undef GO
  return SFM_ERR_ARG;
}

extern "C" int sfm_gemm16_train(const void* A, const void* W, const float* bias, void* out, const float* resid,
                                float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                                long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                                int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                                int nsplit, int dtype, int variant, float p_drop, unsigned int seed, void* stream) {
  if (epi == EPI_SWISH_DUAL || epi == EPI_SWISH_BWD) return SFM_ERR_ARG;       // those need sfm_gemm16_swish
  return gemm16_impl(A, W, bias, out, resid, gn_partial, B, Lout, Lin, Cin, lda, ksize, stride, pad, a_batch_stride, Kpad, N,
                     Npad, ldo, o_batch_stride, ldr, r_batch_stride, alpha, epi, out_f32, gn_group, nsplit, dtype, variant,
                     p_drop, seed, nullptr, nullptr, stream);
}

// Linear with the FFN's Swish (+ hidden dropout) fused into the epilogue (training; models/conformer.py:44-46 and its backward):
//   backward == 0: z = A W^T + bias; out 16-bit = keep/(1-p) * swish(z), out2 [M, N] 16-bit = d = keep/(1-p) * swish'(z) (saved)
//   backward != 0: out 16-bit = (A W^T) * aux,  aux = the saved d                                (counters: m * N + n, as sfm_ew_train)
extern "C" int sfm_gemm16_swish(const void* A, const void* W, const float* bias, void* out, const void* aux, void* out2, int M,
                                int Cin, int lda, int Kpad, int N, int Npad, int ldo, int backward, float p_drop,
                                unsigned int seed, int dtype, void* stream) {
  // 128 x 128 tiles with the whole-tile epilogue (variant 2): 0.34 / 0.29 ms per launch at M 205 056, N 1024, K 256 against 0.47 /
  // 0.39 ms on the 256-row tiles with the strip epilogue and 0.46 / 0.49 ms on the persistent kernel (tools/gemm_train_bench.py,
  // round 3).  A/B knob: SFM_SWISH_VARIANT = 0 (256-row tiles when they fill the chip), 6, 9, 10.
  const int sv = getenv("SFM_SWISH_VARIANT") ? atoi(getenv("SFM_SWISH_VARIANT")) : 2;   // (read per call: tools/swish_sensitivity.py toggles it)
  return gemm16_impl(A, W, bias, out, nullptr, nullptr, 1, M, M, Cin, lda, 1, 1, 0, 0, Kpad, N, Npad, ldo, 0, 0, 0, 1.0f,
                     backward ? EPI_SWISH_BWD : EPI_SWISH_DUAL, 0, 0, 0, dtype, sv, p_drop, seed, aux, out2, stream);
}

extern "C" int sfm_gemm16_ex(const void* A, const void* W, const float* bias, void* out, const float* resid,
                             float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                             long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                             int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                             int nsplit, int dtype, int variant, void* stream) {
  return sfm_gemm16_train(A, W, bias, out, resid, gn_partial, B, Lout, Lin, Cin, lda, ksize, stride, pad, a_batch_stride,
                          Kpad, N, Npad, ldo, o_batch_stride, ldr, r_batch_stride, alpha, epi, out_f32, gn_group, nsplit,
                          dtype, variant, 0.f, 0u, stream);
}

extern "C" int sfm_gemm16(const void* A, const void* W, const float* bias, void* out, const float* resid,
                          float* gn_partial, int B, int Lout, int Lin, int Cin, int lda, int ksize, int stride, int pad,
                          long long a_batch_stride, int Kpad, int N, int Npad, int ldo, long long o_batch_stride,
                          int ldr, long long r_batch_stride, float alpha, int epi, int out_f32, int gn_group,
                          int nsplit, int dtype, void* stream) {
  return sfm_gemm16_ex(A, W, bias, out, resid, gn_partial, B, Lout, Lin, Cin, lda, ksize, stride, pad, a_batch_stride,
                       Kpad, N, Npad, ldo, o_batch_stride, ldr, r_batch_stride, alpha, epi, out_f32, gn_group, nsplit,
                       dtype, 0, stream);
}
