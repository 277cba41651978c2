/*
 * mdconv_oracle.c -- CPU restatement (plain C + OpenMP) of the reference's hot path:
 * DeformConv2d / ModulatedDeformConv2d / DeformConv3d / ModulatedDeformConv3d forward and
 * backward.  TEST INFRASTRUCTURE ONLY -- see mdconv_oracle.h for who may load it and for the
 * parity-pin statement (PARITY UNPINNED against reference-executed vectors).
 *
 * The arithmetic lives in mdconv_oracle_body.h, instantiated for float and double.
 */
#include "mdconv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/config.h:18 -- EPS is FLT_EPSILON for every dtype */
#define ORACLE_EPS 1.192092896e-07F

/* GET_STEP = gcd by Euclid, src/config.h:43-60 */
static int oracle_gcd(int batch, int step) {
  int mx = batch > step ? batch : step;
  int mn = batch > step ? step : batch;
  while (mx % mn != 0) {
    int t = mx % mn;
    mx = mn;
    mn = t;
  }
  return mn;
}

int oracle_out_size(const oracle_desc *d, int axis) {
  return (d->in_sz[axis] + 2 * d->pad[axis] - (d->dil[axis] * (d->k_sz[axis] - 1) + 1)) /
             d->stride[axis] + 1;
}

/* Shape checks of the host drivers (mdeformable_conv.cu:143-148) plus the ones the reference
 * forgets (divisibility, positive in_step). */
static int oracle_check(const oracle_desc *d, int *nd, int *modulated, int *osz) {
  if (d->op < 0 || d->op > 3) return -1;
  *nd = (d->op == ORACLE_DCN3D || d->op == ORACLE_MDCN3D) ? 3 : 2;
  *modulated = (d->op == ORACLE_MDCN2D || d->op == ORACLE_MDCN3D);
  if (d->batch <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->groups <= 0 || d->dgroups <= 0)
    return -1;
  if (d->in_step <= 0) return -1;
  if (d->c_in % d->groups || d->c_out % d->groups || d->c_in % d->dgroups) return -1;
  if (*nd == 2 && (d->in_sz[2] != 1 || d->k_sz[2] != 1 || d->stride[2] != 1 || d->pad[2] != 0 ||
                   d->dil[2] != 1))
    return -1;
  for (int a = 0; a < 3; ++a) {
    if (d->in_sz[a] <= 0 || d->k_sz[a] <= 0 || d->stride[a] <= 0 || d->dil[a] <= 0 ||
        d->pad[a] < 0)
      return -1;
    osz[a] = oracle_out_size(d, a);
    if (osz[a] <= 0) return -1;
  }
  return 0;
}

/* Storage type of the reference's two intermediate buffers.  `columns` and `grad_columns` are tensors of the INPUT's type
 * (`at::zeros({...}, input.options())`, mdeformable_conv.cu:396-397; 3-D mdeformable_conv3d.cu:494-497), so with half
 * tensors the reference rounds GEMM-1's result before its gradient kernel reads it (:418 -> :425) and the column values
 * before GEMM-2 (:316 -> :436).  Mode 0 (default) keeps them in REAL; 1 / 2 round them to fp16 / bf16 (round to nearest
 * even) at exactly those two points, everything else staying in REAL -- the numerics of the product's native 16-bit
 * kernels, which keep coordinates and accumulators in fp32 (the reference's half COORDINATES are not restated, SURVEY.md
 * section 7).  Used by the tests that pin where the 16-bit kernels' rounding sits. */
static int oracle_inter_mode = 0;
void oracle_set_intermediate_rounding(int mode) { oracle_inter_mode = (mode == 1 || mode == 2) ? mode : 0; }

static float oracle_round_bf16(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return v; /* NaN */
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  memcpy(&v, &u, 4);
  return v;
}
static float oracle_round_f16(float v) {
  /* round to nearest even onto the fp16 grid (normals, subnormals, overflow to inf), via scaling: exact in double */
  if (!(v == v) || v == 0.0f) return v;
  const double a = fabs((double)v);
  if (a >= 65520.0) return v > 0 ? INFINITY : -INFINITY;
  int e;
  (void)frexp(a, &e);                      /* a = m * 2^e, m in [0.5, 1) */
  int ulp_exp = e - 11;                    /* 11 significant bits */
  if (ulp_exp < -24) ulp_exp = -24;        /* subnormal spacing 2^-24 */
  const double q = ldexp(a, -ulp_exp);
  const double r = nearbyint(q);           /* default rounding mode: to nearest even */
  const double res = ldexp(r, ulp_exp);
  return (float)(v > 0 ? res : -res);
}
static double oracle_round_inter(double v) {
  if (oracle_inter_mode == 1) return (double)oracle_round_f16((float)v);
  if (oracle_inter_mode == 2) return (double)oracle_round_bf16((float)v);
  return v;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#define REAL float
#define FN(x) x##_f32
#include "mdconv_oracle_body.h"
#undef REAL
#undef FN

#define REAL double
#define FN(x) x##_f64
#include "mdconv_oracle_body.h"
#undef REAL
#undef FN
