/*
 * mdconv.h -- C ABI of the MI355X-native (gfx950) deformable-convolution library
 * (libmdconv_hip.so).
 *
 * This is the drop-in boundary.  The reference's boundary is the CPython extension module
 * `MDCONV_CUDA` (reference setup.py:37, imported at modulated_deform_conv.py:7) whose eight free
 * functions take at::Tensor handles.  Each entry point below replaces exactly one of those
 * eight; the tensor handles become plain device pointers, the shapes travel in `mdconv_desc`.
 * The Python-side binding that re-creates the `MDCONV_CUDA` module on top of this ABI is
 * modulated_deform_conv_amd/MDCONV_CUDA.py (ctypes); INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - All tensors are contiguous, row-major, on the current HIP device, in the layouts of the
 *    reference:
 *      input  [B, C_in, H, W(, L)]           weight [C_out, C_in/groups, kh, kw(, kl)]
 *      bias   [C_out] (ignored when !with_bias)  output [B, C_out, Ho, Wo(, Lo)]
 *      offset [B, DG*nd*K, Ho, Wo(, Lo)], channel = dg*nd*K + nd*tap + axis, axis order (h, w[, l])
 *      mask   [B, DG*K,    Ho, Wo(, Lo)], channel = dg*K + tap,   tap = (i*kw + j)[*kl + k]
 *  - `stream` is a hipStream_t (NULL = the null stream).  Calls are asynchronous on it.
 *  - `workspace` is caller-owned device scratch of at least mdconv_workspace_bytes() bytes,
 *    16-byte aligned; it may be NULL when that function returns 0.
 *  - Backward entry points ACCUMULATE into every grad_* buffer, which is what the reference's
 *    caller-allocated entry points do (deformable_conv.cu:327-333; the Python wrapper zero-fills,
 *    modulated_deform_conv.py:53-56).  For the modulated-2D op, whose reference entry point
 *    allocates zeros itself (mdeformable_conv.cu:404-411), the binding passes uninitialised
 *    buffers and asks for overwrite mode in the descriptor (`accumulate = 0`, ABI v2).
 *  - `in_step` is accepted for signature parity (reference README.md:30-31) and validated
 *    (> 0); results never depend on it (the reference's own modulated-2D op is in_step-invariant).
 *  - Return value: 0 on success, a negative MDCONV_E* code otherwise; mdconv_last_error() gives
 *    the message for the calling thread.  Unlike the reference (which printf()s and swallows
 *    launch errors, mdeformable_conv.cu:113-117) kernel launch failures are reported.
 */
#ifndef MDCONV_H_
#define MDCONV_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  v1: the eight entry points, call modes as thread-local / process-wide setters
 * (mdconv_set_accumulate, mdconv_set_input_layout, mdconv_set_path).  v2 (this header): the call
 * modes travel IN THE DESCRIPTOR (`accumulate`, `input_layout`, `path`), so a C caller has no hidden
 * state between its calls.  Binary compatibility: a v2 caller ORs MDCONV_DESC_V2 into `ndim`; the
 * library reads the v2 tail of the struct only then.  A descriptor without the tag (a caller built
 * against the v1 header, whose struct ends at `with_bias`) keeps v1 behaviour: the setters apply. */
#define MDCONV_ABI_VERSION 2
#define MDCONV_DESC_V2 0x100   /* flag in mdconv_desc.ndim: the descriptor carries the v2 fields */

/* AT_DISPATCH_FLOATING_TYPES_AND_HALF (mdeformable_conv.cu:101) + bfloat16 */
enum { MDCONV_F32 = 0, MDCONV_F16 = 1, MDCONV_F64 = 2, MDCONV_BF16 = 3 };

enum {
  MDCONV_OK = 0,
  MDCONV_EINVAL = -1,    /* bad descriptor / shape mismatch (reference: AT_ERROR -> RuntimeError) */
  MDCONV_ENULL = -2,     /* a required pointer is NULL */
  MDCONV_EWORKSPACE = -3,/* workspace too small / misaligned */
  MDCONV_ELAUNCH = -4,   /* HIP launch or runtime error */
  MDCONV_EUNSUPPORTED = -5
};

/* Kernel-path selector (tests and benchmarks): AUTO picks the MFMA implicit-GEMM kernels when
 * the shape qualifies and the direct (VALU) kernels otherwise. */
enum { MDCONV_PATH_AUTO = 0, MDCONV_PATH_DIRECT = 1, MDCONV_PATH_MFMA = 2 };

typedef struct mdconv_desc {
  int ndim;       /* 2 or 3, | MDCONV_DESC_V2 when the v2 fields below are filled in */
  int modulated;  /* 0 = DeformConv (DCNv1), 1 = ModulatedDeformConv (DCNv2) */
  int dtype;      /* MDCONV_F32 / F16 / F64 / BF16 -- element type of every tensor */
  int batch;      /* B */
  int c_in;       /* C_in  */
  int c_out;      /* C_out */
  int in_sz[3];   /* H, W, L   (L = 1 when ndim == 2) */
  int k_sz[3];    /* kh, kw, kl (kl = 1 when ndim == 2) */
  int stride[3];  /* (…, 1)  */
  int pad[3];     /* (…, 0)  */
  int dil[3];     /* (…, 1)  */
  int groups;     /* `group` of the reference signature */
  int dgroups;    /* `deformable_group` */
  int in_step;    /* accepted, validated > 0, otherwise unused */
  int with_bias;
  /* ---- ABI v2: read only when ndim carries MDCONV_DESC_V2 (use MDCONV_DESC_INIT) ---- */
  int accumulate;   /* backward write mode: 1 = ACCUMULATE into grad_* (the reference's caller-allocated
                       entry points), 0 = OVERWRITE grad_* (buffers need not be initialised) */
  int input_layout; /* MDCONV_LAYOUT_NCHW or MDCONV_LAYOUT_CHANNELS_LAST (see below) */
  int path;         /* MDCONV_PATH_AUTO = the process default (MDCONV_PATH / mdconv_set_path), or a forced
                       MDCONV_PATH_DIRECT / MDCONV_PATH_MFMA for this call */
  int reserved[5];  /* must be 0 */
} mdconv_desc;

/* Initialiser of a v2 descriptor: `mdconv_desc d = MDCONV_DESC_INIT(2);` then fill in the shape.
 * (reference semantics by default: accumulate, NCHW input, process-default path) */
#define MDCONV_DESC_INIT(nd) { (nd) | MDCONV_DESC_V2, 0, 0, 0, 0, 0, {0, 0, 1}, {0, 0, 1}, {1, 1, 1}, \
                               {0, 0, 0}, {1, 1, 1}, 1, 1, 64, 0, 1, 0, 0, {0, 0, 0, 0, 0} }

int mdconv_abi_version(void);
const char *mdconv_last_error(void);

/* Output extent on `axis`: (n + 2p - (d(k-1)+1))/s + 1   (mdeformable_conv.cu:150-153). */
int mdconv_out_size(const mdconv_desc *d, int axis);

/* Scratch bytes needed by the forward (backward = 0) or backward (backward = 1) of `d`. */
size_t mdconv_workspace_bytes(const mdconv_desc *d, int backward);

/* Process-wide default kernel path (MDCONV_PATH_*) for descriptors that do not name one; returns the
 * previous value.  The environment variable MDCONV_PATH=auto|direct|mfma sets the initial default.
 * Per call: mdconv_desc.path (ABI v2). */
int mdconv_set_path(int path);
/* Path the last forward / backward call of this thread actually ran (MDCONV_PATH_DIRECT/MFMA). */
int mdconv_last_path(void);
/* Kernel family behind it: the shape-generic VALU kernels, the fp32 MFMA kernels (also used for
 * 16-bit tensors through fp32 copies when the native kernels do not cover the shape), or the
 * native fp16 / bf16 MFMA kernels. */
enum { MDCONV_KERNELS_DIRECT = 1, MDCONV_KERNELS_F32 = 2, MDCONV_KERNELS_HP = 3 };
int mdconv_last_kernels(void);

/* Per-kernel timing for benchmarks: when enabled, the four dominant kernels are bracketed by HIP
 * events ON THE CALLER'S STREAM.  After a stream/device synchronise, mdconv_profile_read() returns
 * the number of launches of kernel `which` (0 = forward GEMM, 1 = backward data GEMM [the fused
 * backward kernel of the 16-bit path], 2 = backward weight GEMM, 3 = grad_input gather, 4 = coordinate gradients [fp32 split drain]) recorded
 * since the last reset and their total duration in ms; mdconv_profile_name() the name of the kernel
 * variant that ran in that slot last (as rocprofv3 prints it, without template arguments). */
int mdconv_profile_enable(int on);
int mdconv_profile_read(int which, double *total_ms);
const char *mdconv_profile_name(int which);
void mdconv_profile_reset(void);

/* ABI v1 setter, kept for callers built against the v1 header: backward write mode of the calling
 * thread for descriptors WITHOUT MDCONV_DESC_V2 (1 = accumulate, the default; 0 = overwrite).  Returns
 * the previous mode.  v2 descriptors carry `accumulate` themselves and ignore it. */
int mdconv_set_accumulate(int on);

/* Memory format of `input`, mdconv_desc.input_layout:
 * MDCONV_LAYOUT_NCHW (default, the reference's [B, C, spatial...]) or MDCONV_LAYOUT_CHANNELS_LAST
 * ([B, spatial..., C], torch.channels_last / channels_last_3d).  Channels-last input is what the
 * native 16-bit kernels gather from, so it saves their layout pass; it is accepted for fp16 / bf16
 * tensors with C_in a multiple of 32 only (MDCONV_EUNSUPPORTED otherwise).  Every other tensor,
 * grad_input included, keeps the reference layout.
 * mdconv_set_input_layout() is the ABI v1 setter (calling thread, descriptors without MDCONV_DESC_V2);
 * it returns the previous setting. */
enum { MDCONV_LAYOUT_NCHW = 0, MDCONV_LAYOUT_CHANNELS_LAST = 1 };
int mdconv_set_input_layout(int layout);
/* 1 if the forward (backward = 0) / backward (backward = 1) of `d` accepts `input` in `layout`,
 * else 0.  The two directions differ (the native 16-bit backward covers fewer shapes than the
 * forward), so a caller that saved a channels-last input for its backward asks here and makes a
 * contiguous copy when the answer is 0 (modulated_deform_conv_amd/MDCONV_CUDA.py does). */
int mdconv_input_layout_supported(const mdconv_desc *d, int layout, int backward);

/* Multi-GPU overlap: every backward records an event on its stream as soon
 * as grad_weight and grad_bias are final -- before the grad_input gather is enqueued.
 * mdconv_stream_wait_weight_ready_on() makes `stream` (a hipStream_t, e.g. the communication
 * stream) wait for that point of the LAST backward issued on `producer_stream` of the current
 * device, whichever host thread issued it (PyTorch runs autograd backwards on worker threads), so
 * the all-reduce of grad_weight || grad_bias runs under the rest of the backward.
 * mdconv_stream_wait_weight_ready() is the stream-less form: the most recent backward on the
 * current device (use the keyed form when several streams run backwards concurrently).
 * Both return MDCONV_EINVAL if no such backward has been issued.  The library keeps one event per
 * (device, stream handle) for the most recently used streams (64 per process; older entries are
 * destroyed).  A stream handle the runtime recycles after hipStreamDestroy matches the event of the
 * destroyed stream until the first backward on the new one: wait only for backwards you issued. */
int mdconv_stream_wait_weight_ready(void *stream);
int mdconv_stream_wait_weight_ready_on(void *stream, void *producer_stream);

/* --- replaces deform_conv2d_forward_cuda (deformable_conv.cu:117-123) ---------------------- */
int mdconv_deform_conv2d_forward(const mdconv_desc *d, const void *input, const void *weight,
                                 const void *bias, const void *offset, void *output,
                                 void *workspace, size_t workspace_bytes, void *stream);
/* --- replaces deform_conv2d_backward_cuda (deformable_conv.cu:327-333) --------------------- */
int mdconv_deform_conv2d_backward(const mdconv_desc *d, const void *input, const void *weight,
                                  const void *bias, const void *offset, void *grad_input,
                                  void *grad_weight, void *grad_bias, void *grad_offset,
                                  const void *grad_output, void *workspace,
                                  size_t workspace_bytes, void *stream);
/* --- replaces modulated_deform_conv2d_forward_cuda (mdeformable_conv.cu:120-126) ----------- */
int mdconv_modulated_deform_conv2d_forward(const mdconv_desc *d, const void *input,
                                           const void *weight, const void *bias,
                                           const void *offset, const void *mask, void *output,
                                           void *workspace, size_t workspace_bytes, void *stream);
/* --- replaces modulated_deform_conv2d_backward_cuda (mdeformable_conv.cu:361-366) ---------- */
int mdconv_modulated_deform_conv2d_backward(const mdconv_desc *d, const void *input,
                                            const void *weight, const void *bias,
                                            const void *offset, const void *mask,
                                            const void *grad_output, void *grad_input,
                                            void *grad_offset, void *grad_mask, void *grad_weight,
                                            void *grad_bias, void *workspace,
                                            size_t workspace_bytes, void *stream);
/* --- replaces deform_conv3d_forward_cuda (deformable_conv3d.cu:160-167) -------------------- */
int mdconv_deform_conv3d_forward(const mdconv_desc *d, const void *input, const void *weight,
                                 const void *bias, const void *offset, void *output,
                                 void *workspace, size_t workspace_bytes, void *stream);
/* --- replaces deform_conv3d_backward_cuda (deformable_conv3d.cu:434-442) ------------------- */
int mdconv_deform_conv3d_backward(const mdconv_desc *d, const void *input, const void *weight,
                                  const void *bias, const void *offset, void *grad_input,
                                  void *grad_weight, void *grad_bias, void *grad_offset,
                                  const void *grad_output, void *workspace,
                                  size_t workspace_bytes, void *stream);
/* --- replaces modulated_deform_conv3d_forward_cuda (mdeformable_conv3d.cu:170-177) --------- */
int mdconv_modulated_deform_conv3d_forward(const mdconv_desc *d, const void *input,
                                           const void *weight, const void *bias,
                                           const void *offset, const void *mask, void *output,
                                           void *workspace, size_t workspace_bytes, void *stream);
/* --- replaces modulated_deform_conv3d_backward_cuda (mdeformable_conv3d.cu:443-451) -------- */
int mdconv_modulated_deform_conv3d_backward(const mdconv_desc *d, const void *input,
                                            const void *weight, const void *bias,
                                            const void *offset, const void *mask,
                                            void *grad_input, void *grad_weight, void *grad_bias,
                                            void *grad_offset, void *grad_mask,
                                            const void *grad_output, void *workspace,
                                            size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDCONV_H_ */
