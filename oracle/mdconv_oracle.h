/*
 * mdconv_oracle.h -- CPU restatement of the reference's deformable-convolution hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library; the product path (modulated_deform_conv_amd/)
 * never links, imports or calls it.
 *
 * PARITY PIN STATUS: the reference (CHONSPQX/modulated-deform-conv v1.0.2) cannot be built in
 * this image (needs nvcc and <THC/THCAtomics.cuh>, src/config.h:11, absent from torch 2.10) and
 * its Python wrapper refuses CPU tensors (modulated_deform_conv.py:22-23), so no reference-executed
 * golden vectors exist.  The reference's only test (my_test.py:1-35) asserts nothing.  The oracle
 * is therefore pinned by (1) the my_test.py scenario's known answers,
 * (2) the zero-offset == F.conv2d/conv3d identity, (3) fp64 central finite differences of every
 * gradient, (4) an independent pure-PyTorch autograd restatement (tests/torch_ref.py).
 * Against reference-executed vectors: PARITY UNPINNED.
 */
#ifndef MDCONV_ORACLE_H_
#define MDCONV_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Which of the reference's four translation units is being restated; they differ in small
 * gating details. */
enum {
  ORACLE_DCN2D = 0,  /* src/deformable_conv.cu    */
  ORACLE_MDCN2D = 1, /* src/mdeformable_conv.cu   */
  ORACLE_DCN3D = 2,  /* src/deformable_conv3d.cu  */
  ORACLE_MDCN3D = 3  /* src/mdeformable_conv3d.cu */
};

typedef struct {
  int op;        /* ORACLE_* */
  int batch;     /* B */
  int c_in;      /* I */
  int c_out;     /* O */
  int in_sz[3];  /* H, W, L (L = 1 for 2-D) */
  int k_sz[3];   /* kh, kw, kl (kl = 1 for 2-D) */
  int stride[3];
  int pad[3];
  int dil[3];
  int groups;    /* G  */
  int dgroups;   /* DG */
  int in_step;   /* chunk = gcd(B, in_step), src/config.h:43-60 */
  int with_bias;
} oracle_desc;

/* Output extent per axis: (n + 2p - (d(k-1)+1))/s + 1  (mdeformable_conv.cu:150-153). */
int oracle_out_size(const oracle_desc *d, int axis);

/* Forward.  All tensors contiguous, layouts of SURVEY.md section 8a.  `mask` may be NULL for the
 * non-modulated ops.  `output` is overwritten.  Returns 0, or -1 on a shape error. */
int oracle_forward_f32(const oracle_desc *d, const float *input, const float *weight,
                       const float *bias, const float *offset, const float *mask, float *output);
int oracle_forward_f64(const oracle_desc *d, const double *input, const double *weight,
                       const double *bias, const double *offset, const double *mask,
                       double *output);

/* Backward.  Every grad_* buffer is ACCUMULATED INTO (the reference's in-place entry points do
 * that, deformable_conv.cu:327-333; its modulated-2D entry point starts from zeros,
 * mdeformable_conv.cu:404-411) -- callers zero them.  grad_mask / mask may be NULL for the
 * non-modulated ops, grad_bias may be NULL when !with_bias. */
int oracle_backward_f32(const oracle_desc *d, const float *input, const float *weight,
                        const float *offset, const float *mask, const float *grad_output,
                        float *grad_input, float *grad_weight, float *grad_bias,
                        float *grad_offset, float *grad_mask);
int oracle_backward_f64(const oracle_desc *d, const double *input, const double *weight,
                        const double *offset, const double *mask, const double *grad_output,
                        double *grad_input, double *grad_weight, double *grad_bias,
                        double *grad_offset, double *grad_mask);

/* Storage type of the reference's `columns` / `grad_columns` buffers (tensors of the input's type, mdeformable_conv.cu:396-397):
 * 0 = as computed (default), 1 = rounded to fp16, 2 = to bf16 -- see mdconv_oracle.c.  Process-wide; tests set and reset it. */
void oracle_set_intermediate_rounding(int mode);

/* Threads the GEMM / per-image loops will use (OpenMP), for bench.py's cpu_baseline.cores. */
int oracle_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
