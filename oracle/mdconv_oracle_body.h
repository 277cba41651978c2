/*
 * mdconv_oracle_body.h -- type-generic body of the oracle; included twice by mdconv_oracle.c with
 * REAL = float / double and FN(x) = x##_f32 / x##_f64.  TEST INFRASTRUCTURE ONLY (see
 * mdconv_oracle.h).  Every block names the reference file:line it restates.  Structure follows
 * the reference: im2col -> per-group GEMM (forward); GEMM-1 -> per-sample gradient loop ->
 * GEMM-2 (backward), chunked by step = gcd(B, in_step).  Index walks are serial in the order of
 * the reference's thread index, so results are deterministic.
 */

/* ------------------------------------------------------------------------------------------
 * GEMMs: stand-ins for ATen addmm_ (mdeformable_conv.cu:180-181, 418, 437-439); plain fp
 * matmul, summation order unspecified by the reference.
 * ---------------------------------------------------------------------------------------- */
/* C[M][N] = A[M][K] . B[K][N] */
static void FN(gemm_nn)(int M, int N, int K, const REAL *A, const REAL *B, REAL *C) {
#pragma omp parallel for schedule(static)
  for (int i0 = 0; i0 < M; i0 += 4) {
    int mi = M - i0 < 4 ? M - i0 : 4;
    for (int r = 0; r < mi; ++r)
      for (int j = 0; j < N; ++j) C[(size_t)(i0 + r) * N + j] = 0;
    if (mi == 4) {
      REAL *c0 = C + (size_t)i0 * N, *c1 = c0 + N, *c2 = c1 + N, *c3 = c2 + N;
      for (int k = 0; k < K; ++k) {
        const REAL a0 = A[(size_t)i0 * K + k], a1 = A[(size_t)(i0 + 1) * K + k];
        const REAL a2 = A[(size_t)(i0 + 2) * K + k], a3 = A[(size_t)(i0 + 3) * K + k];
        const REAL *b = B + (size_t)k * N;
        for (int j = 0; j < N; ++j) {
          const REAL bv = b[j];
          c0[j] += a0 * bv; c1[j] += a1 * bv; c2[j] += a2 * bv; c3[j] += a3 * bv;
        }
      }
    } else {
      for (int r = 0; r < mi; ++r) {
        REAL *c = C + (size_t)(i0 + r) * N;
        for (int k = 0; k < K; ++k) {
          const REAL a = A[(size_t)(i0 + r) * K + k];
          const REAL *b = B + (size_t)k * N;
          for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
      }
    }
  }
}

/* C[M][N] = A[K][M]^T . B[K][N] */
static void FN(gemm_tn)(int M, int N, int K, const REAL *A, const REAL *B, REAL *C) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    REAL *c = C + (size_t)i * N;
    for (int j = 0; j < N; ++j) c[j] = 0;
    for (int k = 0; k < K; ++k) {
      const REAL a = A[(size_t)k * M + i];
      const REAL *b = B + (size_t)k * N;
      for (int j = 0; j < N; ++j) c[j] += a * b[j];
    }
  }
}

/* C[M][N] += A[M][K] . B[N][K]^T */
static void FN(gemm_nt_acc)(int M, int N, int K, const REAL *A, const REAL *B, REAL *C) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      const REAL *a = A + (size_t)i * K, *b = B + (size_t)j * K;
      REAL s = 0;
#pragma omp simd reduction(+ : s)
      for (int k = 0; k < K; ++k) s += a[k] * b[k];
      C[(size_t)i * N + j] += s;
    }
}

/* ------------------------------------------------------------------------------------------
 * Forward sampler: *_im2col_bilinear (mdeformable_conv.cu:4-35, deformable_conv.cu:3-34) and
 * *_im2col_trilinear (deformable_conv3d.cu:3-52, mdeformable_conv3d.cu:3-52).
 * Corner ci: bit (nd-1-a) set <=> axis a takes the `high` neighbour, which reproduces the
 * reference's v1..v4 / v1..v8 order; weights are multiplied h-first, summed in ci order.
 * A `low` corner is read iff low >= 0, a `high` corner iff high <= size-1 (the caller's range
 * gate makes the other two inequalities true).
 * ---------------------------------------------------------------------------------------- */
static REAL FN(sample_fwd)(const REAL *plane, int nd, const int *sz, const REAL *p) {
  int low[3];
  REAL l[3];
  for (int a = 0; a < nd; ++a) {
    low[a] = (int)floor((double)p[a]);
    l[a] = p[a] - (REAL)low[a];
  }
  REAL val = 0;
  for (int ci = 0; ci < (1 << nd); ++ci) {
    int ok = 1;
    size_t idx = 0;
    REAL w = 1;
    for (int a = 0; a < nd; ++a) {
      const int hi = (ci >> (nd - 1 - a)) & 1;
      const int pos = low[a] + hi;
      if (hi ? (pos > sz[a] - 1) : (pos < 0)) ok = 0;
      idx = idx * (size_t)sz[a] + (size_t)(pos < 0 ? 0 : pos);
      w = (a == 0) ? (hi ? l[a] : (REAL)1 - l[a]) : w * (hi ? l[a] : (REAL)1 - l[a]);
    }
    const REAL v = ok ? plane[idx] : (REAL)0;
    val = (ci == 0) ? w * v : val + w * v;
  }
  return val;
}

/* *_im2col_gpu_kernel: mdeformable_conv.cu:37-87, deformable_conv.cu:36-85,
 * deformable_conv3d.cu:54-119, mdeformable_conv3d.cu:54-127.  One reference thread per
 * (c_im, b_col, out-pixel); loops over the K taps.  columns[(c*K + tap)][b*S_o + pix]. */
static void FN(im2col)(const oracle_desc *d, int nd, int modulated, int step, const int *osz,
                       const REAL *im, const REAL *off, const REAL *msk, REAL *col) {
  const int C = d->c_in, DG = d->dgroups;
  const int K = d->k_sz[0] * d->k_sz[1] * d->k_sz[2];
  const size_t S_i = (size_t)d->in_sz[0] * d->in_sz[1] * d->in_sz[2];
  const size_t S_o = (size_t)osz[0] * osz[1] * osz[2];
  const int cpdg = C / DG; /* channel_per_deformable_group, mdeformable_conv.cu:98 */
#pragma omp parallel for schedule(static) collapse(2)
  for (int c = 0; c < C; ++c)
    for (int b = 0; b < step; ++b) {
      const int dg = c / cpdg;
      const REAL *plane = im + ((size_t)b * C + c) * S_i;
      const REAL *offp = off + ((size_t)b * DG + dg) * (size_t)nd * K * S_o;
      const REAL *mskp = modulated ? msk + ((size_t)b * DG + dg) * (size_t)K * S_o : NULL;
      for (size_t pix = 0; pix < S_o; ++pix) {
        int o[3];
        o[2] = (int)(pix % (size_t)osz[2]);
        o[1] = (int)((pix / (size_t)osz[2]) % (size_t)osz[1]);
        o[0] = (int)(pix / ((size_t)osz[2] * osz[1]));
        for (int tap = 0; tap < K; ++tap) {
          int t[3];
          t[2] = tap % d->k_sz[2];
          t[1] = (tap / d->k_sz[2]) % d->k_sz[1];
          t[0] = tap / (d->k_sz[2] * d->k_sz[1]);
          REAL p[3];
          int inside = 1;
          for (int a = 0; a < nd; ++a) {
            const REAL delta = offp[((size_t)nd * tap + a) * S_o + pix];
            /* h_im = h_in + i*dilation_h + offset_h, mdeformable_conv.cu:78-79 */
            p[a] = (REAL)(o[a] * d->stride[a] - d->pad[a] + t[a] * d->dil[a]) + delta;
            if (!(p[a] > (REAL)-1 && p[a] < (REAL)d->in_sz[a])) inside = 0; /* :80 */
          }
          REAL val = 0;
          if (inside) val = FN(sample_fwd)(plane, nd, d->in_sz, p);
          if (modulated) val = val * mskp[(size_t)tap * S_o + pix]; /* :83 */
          col[((size_t)c * K + tap) * ((size_t)step * S_o) + (size_t)b * S_o + pix] = val;
        }
      }
    }
}

/* ------------------------------------------------------------------------------------------
 * Forward host driver: *_forward_cuda (mdeformable_conv.cu:120-194, deformable_conv.cu:117-196,
 * deformable_conv3d.cu:160-256, mdeformable_conv3d.cu:170-262).  Restates the in_step-invariant
 * operator (the reference's R1 permutation bug in three of the four ops is not reproduced;
 * SURVEY.md section 8a R1).
 * ---------------------------------------------------------------------------------------- */
int FN(oracle_forward)(const oracle_desc *d, const REAL *input, const REAL *weight,
                       const REAL *bias, const REAL *offset, const REAL *mask, REAL *output) {
  int nd, modulated, osz[3];
  if (oracle_check(d, &nd, &modulated, osz)) return -1;
  if (modulated && !mask) return -1;
  const int B = d->batch, C = d->c_in, O = d->c_out, G = d->groups, DG = d->dgroups;
  const int K = d->k_sz[0] * d->k_sz[1] * d->k_sz[2];
  const size_t S_i = (size_t)d->in_sz[0] * d->in_sz[1] * d->in_sz[2];
  const size_t S_o = (size_t)osz[0] * osz[1] * osz[2];
  const int step = oracle_gcd(B, d->in_step); /* GET_STEP, config.h:43-60 */
  const size_t ncol = (size_t)step * S_o;
  const int rows_g = C / G * K; /* column rows per conv group, mdeformable_conv.cu:178 */
  const int Og = O / G;
  REAL *columns = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * ncol);
  REAL *outg = (REAL *)malloc(sizeof(REAL) * (size_t)Og * ncol);
  if (!columns || !outg) { free(columns); free(outg); return -1; }
  for (int b = 0; b < B / step; ++b) {
    FN(im2col)(d, nd, modulated, step, osz, input + (size_t)b * step * C * S_i,
               offset + (size_t)b * step * DG * nd * K * S_o,
               modulated ? mask + (size_t)b * step * DG * K * S_o : NULL, columns);
    for (int g = 0; g < G; ++g) {
      /* output[b][g] = W[g].flatten(1) @ columns[g], mdeformable_conv.cu:179-182 */
      FN(gemm_nn)(Og, (int)ncol, rows_g, weight + (size_t)g * Og * rows_g,
                  columns + (size_t)g * rows_g * ncol, outg);
      /* [B/step, O, step, S_o] -> [B, O, S_o] (:187-189) and + bias (:190-192) */
      for (int o = 0; o < Og; ++o)
        for (int s = 0; s < step; ++s) {
          REAL *dst = output + (((size_t)b * step + s) * O + (size_t)g * Og + o) * S_o;
          const REAL *src = outg + (size_t)o * ncol + (size_t)s * S_o;
          const REAL bv = d->with_bias ? bias[g * Og + o] : (REAL)0;
          for (size_t pix = 0; pix < S_o; ++pix) dst[pix] = src[pix] + bv;
        }
    }
  }
  free(columns);
  free(outg);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * *_gradient_gpu_kernel: one reference thread per SAMPLE `index` over
 * [C*K][step*S_o] (mdeformable_conv.cu:202-318, deformable_conv.cu:198-287,
 * deformable_conv3d.cu:259-389, mdeformable_conv3d.cu:265-395).  The four files differ in:
 *   load_eps : `high` corners are LOADED only if d > EPS   (dcn2d :254-261, both 3-D :336-338)
 *   atom_eps : `high` corners are SCATTERED only if d > EPS (mdcn2d :285-293, both 3-D :336-338)
 *   mdcn2d   : grad_input weights written as (low+1-p)/(p+1-high) (:282-293); grad_offset only
 *              inside (-1,size) on every axis and with (low+1-p)/(p-low) factors (:295-314).
 * atomicAdd -> serial +=.
 * ---------------------------------------------------------------------------------------- */
static void FN(gradient_loop)(const oracle_desc *d, int nd, int modulated, int step,
                              const int *osz, const REAL *grad_col, const REAL *im,
                              const REAL *off, const REAL *msk, REAL *columns, REAL *grad_im,
                              REAL *grad_off, REAL *grad_msk) {
  const int C = d->c_in, DG = d->dgroups;
  const int K = d->k_sz[0] * d->k_sz[1] * d->k_sz[2];
  const size_t S_i = (size_t)d->in_sz[0] * d->in_sz[1] * d->in_sz[2];
  const size_t S_o = (size_t)osz[0] * osz[1] * osz[2];
  const int load_eps = d->op != ORACLE_MDCN2D;
  const int atom_eps = d->op != ORACLE_DCN2D;
  const int mdcn2d = d->op == ORACLE_MDCN2D;
  const REAL eps = (REAL)ORACLE_EPS;
  const size_t ncol = (size_t)step * S_o;
  const size_t n = (size_t)C * K * ncol;
  for (size_t index = 0; index < n; ++index) {
    const int tap = (int)((index / ncol) % (size_t)K);
    const int bpos = (int)((index % ncol) / S_o);
    const size_t pix = index % S_o;
    const int cpos_in = (int)(index / ncol / (size_t)K);
    const int dg = cpos_in / (C / DG); /* offset_group_index, mdeformable_conv.cu:231 */
    int o[3], t[3];
    o[2] = (int)(pix % (size_t)osz[2]);
    o[1] = (int)((pix / (size_t)osz[2]) % (size_t)osz[1]);
    o[0] = (int)(pix / ((size_t)osz[2] * osz[1]));
    t[2] = tap % d->k_sz[2];
    t[1] = (tap / d->k_sz[2]) % d->k_sz[1];
    t[0] = tap / (d->k_sz[2] * d->k_sz[1]);
    size_t off_ptr[3];
    for (int a = 0; a < nd; ++a)
      off_ptr[a] = (((size_t)bpos * DG + dg) * (size_t)nd * K + (size_t)nd * tap + a) * S_o + pix;
    const size_t msk_ptr = (((size_t)bpos * DG + dg) * (size_t)K + tap) * S_o + pix;
    REAL p[3], dd[3];
    int low[3], inside = 1;
    for (int a = 0; a < nd; ++a) {
      p[a] = (REAL)(o[a] * d->stride[a] - d->pad[a] + t[a] * d->dil[a]) + off[off_ptr[a]];
      low[a] = (int)floor((double)p[a]);
      dd[a] = p[a] - (REAL)low[a];
      if (!(p[a] > (REAL)-1 && p[a] < (REAL)d->in_sz[a])) inside = 0;
    }
    const REAL m = modulated ? msk[msk_ptr] : (REAL)1;
    const REAL gc = grad_col[index];
    const REAL dval = modulated ? m * gc : gc; /* mdeformable_conv.cu:280 */
    const size_t base = ((size_t)bpos * C + cpos_in) * S_i;
    REAL v[8], w[8];
    REAL val = 0;
    for (int ci = 0; ci < (1 << nd); ++ci) {
      int ok_load = 1, ok_atom = 1;
      size_t idx = 0;
      REAL wt = 1, wt_alt = 1;
      for (int a = 0; a < nd; ++a) {
        const int hi = (ci >> (nd - 1 - a)) & 1;
        const int pos = low[a] + hi;
        const int in_img = pos >= 0 && pos <= d->in_sz[a] - 1;
        if (!in_img) { ok_load = 0; ok_atom = 0; }
        if (hi && !(dd[a] > eps)) {
          if (load_eps) ok_load = 0;
          if (atom_eps) ok_atom = 0;
        }
        idx = idx * (size_t)d->in_sz[a] + (size_t)(in_img ? pos : 0);
        const REAL f = hi ? dd[a] : (REAL)1 - dd[a];
        /* mdcn2d scatter weight: (low+1-p) / (p+1-high), mdeformable_conv.cu:284-293 */
        const REAL f_alt = hi ? (p[a] + (REAL)1 - (REAL)(low[a] + 1)) : ((REAL)(low[a] + 1) - p[a]);
        wt = (a == 0) ? f : wt * f;
        wt_alt = (a == 0) ? f_alt : wt_alt * f_alt;
      }
      v[ci] = ok_load ? im[base + idx] : (REAL)0;
      w[ci] = wt;
      if (ok_atom) grad_im[base + idx] += mdcn2d ? wt_alt * dval : wt * dval;
      val = (ci == 0) ? wt * v[ci] : val + wt * v[ci];
    }
    /* grad_offset: sum over corners of (+/-)(product of the OTHER axes' weights) * v */
    if (!mdcn2d || inside) {
      for (int a = 0; a < nd; ++a) {
        REAL acc = 0;
        for (int ci = 0; ci < (1 << nd); ++ci) {
          const int hi_a = (ci >> (nd - 1 - a)) & 1;
          REAL f = hi_a ? (REAL)1 : (REAL)-1;
          for (int a2 = 0; a2 < nd; ++a2) {
            if (a2 == a) continue;
            const int hi = (ci >> (nd - 1 - a2)) & 1;
            const REAL g = mdcn2d ? (hi ? (p[a2] - (REAL)low[a2]) : ((REAL)(low[a2] + 1) - p[a2]))
                                  : (hi ? dd[a2] : (REAL)1 - dd[a2]);
            f = f * g;
          }
          acc = (ci == 0) ? f * v[ci] : acc + f * v[ci];
        }
        /* mdcn2d: w_tmp*grad_col*mask (:306, :313); others: (...)*dval */
        grad_off[off_ptr[a]] += mdcn2d ? acc * gc * m : acc * dval;
      }
    }
    if (modulated) grad_msk[msk_ptr] += gc * val; /* :315 / mdeformable_conv3d.cu:392 */
    columns[index] = modulated ? val * m : val;    /* :316 */
    (void)w;
  }
}

/* ------------------------------------------------------------------------------------------
 * Backward host driver: *_backward_cuda (mdeformable_conv.cu:361-458, deformable_conv.cu:327-431,
 * deformable_conv3d.cu:434-561, mdeformable_conv3d.cu:443-586).
 * ---------------------------------------------------------------------------------------- */
int FN(oracle_backward)(const oracle_desc *d, const REAL *input, const REAL *weight,
                        const REAL *offset, const REAL *mask, const REAL *grad_output,
                        REAL *grad_input, REAL *grad_weight, REAL *grad_bias, REAL *grad_offset,
                        REAL *grad_mask) {
  int nd, modulated, osz[3];
  if (oracle_check(d, &nd, &modulated, osz)) return -1;
  if (modulated && (!mask || !grad_mask)) return -1;
  if (d->with_bias && !grad_bias) return -1;
  const int B = d->batch, C = d->c_in, O = d->c_out, G = d->groups, DG = d->dgroups;
  const int K = d->k_sz[0] * d->k_sz[1] * d->k_sz[2];
  const size_t S_i = (size_t)d->in_sz[0] * d->in_sz[1] * d->in_sz[2];
  const size_t S_o = (size_t)osz[0] * osz[1] * osz[2];
  const int step = oracle_gcd(B, d->in_step);
  const size_t ncol = (size_t)step * S_o;
  const int rows_g = C / G * K, Og = O / G;
  REAL *columns = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * ncol);
  REAL *grad_columns = (REAL *)malloc(sizeof(REAL) * (size_t)C * K * ncol);
  REAL *gout = (REAL *)malloc(sizeof(REAL) * (size_t)O * ncol);
  if (!columns || !grad_columns || !gout) {
    free(columns); free(grad_columns); free(gout);
    return -1;
  }
  for (int b = 0; b < B / step; ++b) {
    /* grad_output -> [B/step, G, O/G, step, S_o] (mdeformable_conv.cu:399-402) */
    for (int oc = 0; oc < O; ++oc)
      for (int s = 0; s < step; ++s)
        memcpy(gout + (size_t)oc * ncol + (size_t)s * S_o,
               grad_output + (((size_t)b * step + s) * O + oc) * S_o, sizeof(REAL) * S_o);
    /* GEMM-1: grad_columns[g] = W[g]^T @ grad_output[b][g]  (beta = 0, :417-419) */
    for (int g = 0; g < G; ++g)
      FN(gemm_tn)(rows_g, (int)ncol, Og, weight + (size_t)g * Og * rows_g,
                  gout + (size_t)g * Og * ncol, grad_columns + (size_t)g * rows_g * ncol);
    if (oracle_inter_mode) { /* grad_columns is a tensor of the input's type (mdeformable_conv.cu:397, :418) */
      const size_t ngc = (size_t)C * K * ncol;
      for (size_t i = 0; i < ngc; ++i) grad_columns[i] = (REAL)oracle_round_inter((double)grad_columns[i]);
    }
    FN(gradient_loop)(d, nd, modulated, step, osz, grad_columns,
                      input + (size_t)b * step * C * S_i,
                      offset + (size_t)b * step * DG * nd * K * S_o,
                      modulated ? mask + (size_t)b * step * DG * K * S_o : NULL, columns,
                      grad_input + (size_t)b * step * C * S_i,
                      grad_offset + (size_t)b * step * DG * nd * K * S_o,
                      modulated ? grad_mask + (size_t)b * step * DG * K * S_o : NULL);
    if (oracle_inter_mode) { /* ... and so is columns (:396, written at :316, read by the addmm_ at :436) */
      const size_t ncl = (size_t)C * K * ncol;
      for (size_t i = 0; i < ncl; ++i) columns[i] = (REAL)oracle_round_inter((double)columns[i]);
    }
    for (int g = 0; g < G; ++g) {
      /* GEMM-2: grad_weight[g] += grad_output[b][g] @ columns[g]^T  (:436-439) */
      FN(gemm_nt_acc)(Og, rows_g, (int)ncol, gout + (size_t)g * Og * ncol,
                      columns + (size_t)g * rows_g * ncol, grad_weight + (size_t)g * Og * rows_g);
      /* grad_bias[g] += grad_output[b][g] @ ones  (:440-444) */
      if (d->with_bias)
        for (int o = 0; o < Og; ++o) {
          const REAL *row = gout + ((size_t)g * Og + o) * ncol;
          REAL s = 0;
          for (size_t j = 0; j < ncol; ++j) s += row[j];
          grad_bias[g * Og + o] += s;
        }
    }
  }
  free(columns);
  free(grad_columns);
  free(gout);
  return 0;
}
