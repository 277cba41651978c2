// mdconv_api.hip -- the C ABI (include/mdconv.h): descriptor validation, path selection, and the
// eight entry points that replace the reference's MDCONV_CUDA exports
// (mdeformable_conv.cu:460-465 registers two; modulated_deform_conv.py calls all eight:
//  :28, :57, :112, :142, :194, :225, :281, :313).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <iterator>
#include <map>
#include <mutex>
#include <utility>

#include "mdconv_common.hpp"
#include "hp_kernels.hpp"
#include "mfma_kernels.hpp"

namespace mdconv {

static thread_local char g_err[512] = "";
static thread_local int g_last_path = 0;
static thread_local int g_last_kernels = 0;
// ABI v1 call modes (descriptors without MDCONV_DESC_V2); v2 descriptors carry their own
static thread_local int g_accumulate = 1;
static thread_local int g_input_layout = 0;   // MDCONV_LAYOUT_*
// "grad_weight / grad_bias are final" events: one per (device, producer stream), shared by all
// host threads (autograd runs the backward on its own worker thread), plus the most recent one per
// device for the stream-less legacy query
static std::mutex g_wready_mu;
struct WReady { hipEvent_t ev; unsigned long long tick; };
static std::map<std::pair<int, hipStream_t>, WReady> g_wready;
static std::map<int, hipEvent_t> g_wready_latest;
static unsigned long long g_wready_tick = 0;
constexpr size_t kWReadyMax = 64;   // streams remembered per process; least recently used are dropped
static std::atomic<int> g_path{-1};  // -1 = not initialised from the environment yet

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return MDCONV_ELAUNCH;
  }
  return MDCONV_OK;
}

static int current_path() {
  int p = g_path.load(std::memory_order_relaxed);
  if (p < 0) {
    const char *e = getenv("MDCONV_PATH");
    p = MDCONV_PATH_AUTO;
    if (e && !strcmp(e, "direct")) p = MDCONV_PATH_DIRECT;
    if (e && !strcmp(e, "mfma")) p = MDCONV_PATH_MFMA;
    int expect = -1;
    g_path.compare_exchange_strong(expect, p);
    p = g_path.load();
  }
  return p;
}

static int desc_ndim(const mdconv_desc *d) { return d->ndim & ~MDCONV_DESC_V2; }

// Call modes of one call: from the descriptor (ABI v2) or from the v1 setters of the calling thread / process.
struct Modes { int accumulate, input_layout, path; };
static int call_modes(const mdconv_desc *d, Modes *m) {
  if (!(d->ndim & MDCONV_DESC_V2)) {
    m->accumulate = g_accumulate;
    m->input_layout = g_input_layout;
    m->path = current_path();
    return MDCONV_OK;
  }
  if ((d->accumulate != 0 && d->accumulate != 1) ||
      (d->input_layout != MDCONV_LAYOUT_NCHW && d->input_layout != MDCONV_LAYOUT_CHANNELS_LAST) ||
      d->path < MDCONV_PATH_AUTO || d->path > MDCONV_PATH_MFMA) {
    set_error("bad call mode in the descriptor (accumulate=%d, input_layout=%d, path=%d)", d->accumulate,
              d->input_layout, d->path);
    return MDCONV_EINVAL;
  }
  for (int i = 0; i < 5; ++i)
    if (d->reserved[i] != 0) {
      set_error("mdconv_desc.reserved must be 0");
      return MDCONV_EINVAL;
    }
  m->accumulate = d->accumulate;
  m->input_layout = d->input_layout;
  m->path = d->path == MDCONV_PATH_AUTO ? current_path() : d->path;
  return MDCONV_OK;
}

int fill_geom(const mdconv_desc *d, Geom *g) {
  if (!d) {
    set_error("descriptor is NULL");
    return MDCONV_ENULL;
  }
  const int ndim = desc_ndim(d);
  if (ndim != 2 && ndim != 3) {
    set_error("ndim must be 2 or 3 (got %d)", ndim);
    return MDCONV_EINVAL;
  }
  if (d->dtype != MDCONV_F32 && d->dtype != MDCONV_F16 && d->dtype != MDCONV_F64 &&
      d->dtype != MDCONV_BF16) {
    set_error("unsupported dtype %d", d->dtype);
    return MDCONV_EINVAL;
  }
  if (d->batch <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->groups <= 0 || d->dgroups <= 0) {
    set_error("batch/channels/groups must be positive");
    return MDCONV_EINVAL;
  }
  if (d->in_step <= 0) {  // the reference divides by it (config.h:43-60)
    set_error("in_step must be positive (got %d)", d->in_step);
    return MDCONV_EINVAL;
  }
  if (d->c_in % d->groups || d->c_out % d->groups) {
    set_error("Input shape and kernel channels wont match: channels %d / %d not divisible by group %d",
              d->c_in, d->c_out, d->groups);
    return MDCONV_EINVAL;
  }
  if (d->c_in % d->dgroups) {
    set_error("channels %d not divisible by deformable_group %d", d->c_in, d->dgroups);
    return MDCONV_EINVAL;
  }
  memset(g, 0, sizeof(*g));
  g->nd = ndim;
  g->B = d->batch;
  g->C = d->c_in;
  g->O = d->c_out;
  g->G = d->groups;
  g->DG = d->dgroups;
  int64_t S_i = 1, S_o = 1, K = 1;
  for (int a = 0; a < 3; ++a) {
    const bool used = a < ndim;
    const int n = used ? d->in_sz[a] : 1, k = used ? d->k_sz[a] : 1, s = used ? d->stride[a] : 1;
    const int p = used ? d->pad[a] : 0, dl = used ? d->dil[a] : 1;
    if (n <= 0 || k <= 0 || s <= 0 || dl <= 0 || p < 0) {
      set_error("bad size/kernel/stride/dilation/padding on axis %d", a);
      return MDCONV_EINVAL;
    }
    const int o = (n + 2 * p - (dl * (k - 1) + 1)) / s + 1;
    if (n + 2 * p - (dl * (k - 1) + 1) < 0 || o <= 0) {
      set_error("empty output on axis %d", a);
      return MDCONV_EINVAL;
    }
    g->in_sz[a] = n;
    g->ksz[a] = k;
    g->stride[a] = s;
    g->pad[a] = p;
    g->dil[a] = dl;
    g->out_sz[a] = o;
    S_i *= n;
    S_o *= o;
    K *= k;
  }
  const int64_t lim = 0x7fffffffLL;
  if (S_i > lim || S_o * d->batch > lim || K > 4096 ||
      (int64_t)d->batch * d->c_in * S_i > (int64_t)1 << 40) {
    set_error("tensor too large for 32-bit pixel indexing");
    return MDCONV_EUNSUPPORTED;
  }
  g->S_i = (int)S_i;
  g->S_o = (int)S_o;
  g->K = (int)K;
  g->N = (int)(S_o * d->batch);
  g->Cg = d->c_in / d->groups;
  g->Og = d->c_out / d->groups;
  g->Cdg = d->c_in / d->dgroups;
  g->with_bias = d->with_bias ? 1 : 0;
  g->modulated = d->modulated ? 1 : 0;
  g->acc_data = g->acc_w = 1;
  // gating flavours of the four reference files
  const bool mdcn2d = ndim == 2 && d->modulated;
  const bool dcn2d = ndim == 2 && !d->modulated;
  g->load_eps = mdcn2d ? 0 : 1;
  g->atom_eps = dcn2d ? 0 : 1;
  g->range_gate = mdcn2d ? 1 : 0;
  return MDCONV_OK;
}

// One line on stderr, once per process, when a shape with matrix-sized channel counts runs on the
// shape-generic VALU kernels (an order of magnitude slower than the matrix-core kernels): nothing else tells the user
// (mdconv_last_kernels() reports it per call; MDCONV_QUIET=1 silences the line).
static void note_direct_fallback(const Geom &g, int dtype, bool backward, int path) {
  static std::atomic<bool> said{false};
  if (g.Cg < 16 || g.Og < 16 || dtype == MDCONV_F64 || path == MDCONV_PATH_DIRECT) return;
  if (!backward && g.Cg < 64) return;   // narrow conv groups: the shape-generic forward is no slower (mfma_kernels.hip)
  if (said.exchange(true)) return;
  const char *q = getenv("MDCONV_QUIET");
  if (q && atoi(q) != 0) return;
  fprintf(stderr,
          "mdconv: %s of a %d-D shape with C_in=%d C_out=%d groups=%d deformable_groups=%d runs on the shape-generic "
          "kernels (the matrix-core kernels need %s); expect it to be ~10x slower. This note is printed once.\n",
          backward ? "backward" : "forward", g.nd, g.C, g.O, g.G, g.DG,
          "with several deformable groups: one conv group and C_in/deformable_groups of at least 8, or with conv groups "
          "C_in/deformable_groups a multiple of 8, at least 16, aligned with them");
}

static int require(const void *p, const char *name) {
  if (!p) {
    set_error("%s pointer is NULL", name);
    return MDCONV_ENULL;
  }
  return MDCONV_OK;
}

static int check_ws(void *ws, size_t have, size_t need) {
  if (need == 0) return MDCONV_OK;
  if (!ws || have < need) {
    set_error("workspace too small: need %zu bytes, have %zu", need, have);
    return MDCONV_EWORKSPACE;
  }
  if (((uintptr_t)ws & 15) != 0) {
    set_error("workspace must be 16-byte aligned");
    return MDCONV_EWORKSPACE;
  }
  return MDCONV_OK;
}

static int run_forward(const mdconv_desc *d, int nd, int modulated, Tensors t, void *ws,
                       size_t ws_bytes, void *stream) {
  g_err[0] = 0;
  Geom g;
  int rc = fill_geom(d, &g);
  if (rc) return rc;
  if (g.nd != nd || (d->modulated != 0) != (modulated != 0)) {
    set_error("descriptor (ndim=%d, modulated=%d) does not match this entry point", g.nd,
              d->modulated);
    return MDCONV_EINVAL;
  }
  Modes md;
  if ((rc = call_modes(d, &md))) return rc;
  if ((rc = require(t.input, "input")) || (rc = require(t.weight, "weight")) ||
      (rc = require(t.offset, "offset")) || (rc = require(t.output, "output")))
    return rc;
  if (modulated && (rc = require(t.mask, "mask"))) return rc;
  if (g.with_bias && (rc = require(t.bias, "bias"))) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int path = md.path;
  g.in_cl = md.input_layout == MDCONV_LAYOUT_CHANNELS_LAST ? 1 : 0;
  if (g.in_cl && !(path != MDCONV_PATH_DIRECT && hp_supported(g, d->dtype, false) && g.C % 32 == 0)) {
    set_error("channels-last input is only supported by the native 16-bit kernels with C_in a multiple of 32");
    return MDCONV_EUNSUPPORTED;
  }
  // 16-bit tensors: native fp16 / bf16 kernels (hp_*.hip) when the shape qualifies (and is not one of the few-tile forwards
  // that the fp32 kernels run faster: hp_forward_preferred)
  if (path != MDCONV_PATH_DIRECT && hp_supported(g, d->dtype, false) && (g.in_cl || hp_forward_preferred(g, d->dtype))) {
    if ((rc = check_ws(ws, ws_bytes, hp_workspace_bytes(g, d->dtype, false)))) return rc;
    g_last_path = MDCONV_PATH_MFMA;
    g_last_kernels = MDCONV_KERNELS_HP;
    return hp_forward(g, d->dtype, t, ws, s);
  }
  const bool mfma_ok = mfma_supported(g, d->dtype, false);
  if (path == MDCONV_PATH_MFMA && !mfma_ok) {
    set_error("MDCONV_PATH=mfma but this shape/dtype is not supported by the MFMA kernels");
    return MDCONV_EUNSUPPORTED;
  }
  if (mfma_ok && path != MDCONV_PATH_DIRECT) {
    if ((rc = check_ws(ws, ws_bytes, mfma_workspace_bytes(g, d->dtype, false)))) return rc;
    g_last_path = MDCONV_PATH_MFMA;
    g_last_kernels = MDCONV_KERNELS_F32;
    return mfma_forward(g, d->dtype, t, ws, s);
  }
  g_last_path = MDCONV_PATH_DIRECT;
  g_last_kernels = MDCONV_KERNELS_DIRECT;
  note_direct_fallback(g, d->dtype, false, path);
  return direct_forward(g, d->dtype, t, s);
}

static int run_backward(const mdconv_desc *d, int nd, int modulated, Tensors t, void *ws,
                        size_t ws_bytes, void *stream) {
  g_err[0] = 0;
  Geom g;
  int rc = fill_geom(d, &g);
  if (rc) return rc;
  if (g.nd != nd || (d->modulated != 0) != (modulated != 0)) {
    set_error("descriptor (ndim=%d, modulated=%d) does not match this entry point", g.nd,
              d->modulated);
    return MDCONV_EINVAL;
  }
  Modes md;
  if ((rc = call_modes(d, &md))) return rc;
  if ((rc = require(t.input, "input")) || (rc = require(t.weight, "weight")) ||
      (rc = require(t.offset, "offset")) || (rc = require(t.grad_output, "grad_output")) ||
      (rc = require(t.grad_input, "grad_input")) || (rc = require(t.grad_weight, "grad_weight")) ||
      (rc = require(t.grad_offset, "grad_offset")))
    return rc;
  if (modulated && ((rc = require(t.mask, "mask")) || (rc = require(t.grad_mask, "grad_mask"))))
    return rc;
  if (g.with_bias && (rc = require(t.grad_bias, "grad_bias"))) return rc;
  hipStream_t s = (hipStream_t)stream;
  g.acc_data = g.acc_w = md.accumulate;
  const int path = md.path;
  g.in_cl = md.input_layout == MDCONV_LAYOUT_CHANNELS_LAST ? 1 : 0;
  if (g.in_cl && !(path != MDCONV_PATH_DIRECT && hp_supported(g, d->dtype, true) && g.C % 32 == 0)) {
    set_error("channels-last input is only supported by the native 16-bit kernels with C_in a multiple of 32");
    return MDCONV_EUNSUPPORTED;
  }
  if (path != MDCONV_PATH_DIRECT && hp_supported(g, d->dtype, true)) {
    if ((rc = check_ws(ws, ws_bytes, hp_workspace_bytes(g, d->dtype, true)))) return rc;
    g_last_path = MDCONV_PATH_MFMA;
    g_last_kernels = MDCONV_KERNELS_HP;
    return hp_backward(g, d->dtype, t, ws, s);
  }
  const bool mfma_ok = mfma_supported(g, d->dtype, true);
  if (path == MDCONV_PATH_MFMA && !mfma_ok) {
    set_error("MDCONV_PATH=mfma but this shape/dtype is not supported by the MFMA kernels");
    return MDCONV_EUNSUPPORTED;
  }
  if (mfma_ok && path != MDCONV_PATH_DIRECT) {
    if ((rc = check_ws(ws, ws_bytes, mfma_workspace_bytes(g, d->dtype, true)))) return rc;
    g_last_path = MDCONV_PATH_MFMA;
    g_last_kernels = MDCONV_KERNELS_F32;
    return mfma_backward(g, d->dtype, t, ws, s);
  }
  g_last_path = MDCONV_PATH_DIRECT;
  g_last_kernels = MDCONV_KERNELS_DIRECT;
  note_direct_fallback(g, d->dtype, true, path);
  if (d->dtype == MDCONV_F16 || d->dtype == MDCONV_BF16) {
    if ((rc = check_ws(ws, ws_bytes, direct16_workspace_bytes(g)))) return rc;
    if ((rc = direct16_backward(g, d->dtype, t, ws, s))) return rc;
    return record_weight_ready(s);
  }
  if (!md.accumulate) {
    // the direct kernels scatter with atomics, so "overwrite" means: clear first
    const size_t es = d->dtype == MDCONV_F64 ? 8 : (d->dtype == MDCONV_F32 ? 4 : 2);
    const size_t n_off = (size_t)g.B * g.DG * g.nd * g.K * g.S_o, n_m = (size_t)g.B * g.DG * g.K * g.S_o;
    if ((rc = zero_bytes(t.grad_input, (size_t)g.B * g.C * g.S_i * es, s)) ||
        (rc = zero_bytes(t.grad_offset, n_off * es, s)) ||
        (rc = zero_bytes(t.grad_weight, (size_t)g.O * g.Cg * g.K * es, s)))
      return rc;
    if (modulated && (rc = zero_bytes(t.grad_mask, n_m * es, s))) return rc;
    if (g.with_bias && (rc = zero_bytes(t.grad_bias, (size_t)g.O * es, s))) return rc;
  }
  if ((rc = direct_backward(g, d->dtype, t, s))) return rc;
  return record_weight_ready(s);
}

int record_weight_ready(hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MDCONV_ELAUNCH;
  std::lock_guard<std::mutex> lock(g_wready_mu);
  const auto key = std::make_pair(dev, stream);
  auto it = g_wready.find(key);
  if (it == g_wready.end()) {
    if (g_wready.size() >= kWReadyMax) {   // programs that create and destroy many streams: evict the oldest
      auto old = g_wready.begin();
      for (auto j = g_wready.begin(); j != g_wready.end(); ++j)
        if (j->second.tick < old->second.tick) old = j;
      for (auto j = g_wready_latest.begin(); j != g_wready_latest.end();)
        j = (j->second == old->second.ev) ? g_wready_latest.erase(j) : std::next(j);
      (void)hipEventDestroy(old->second.ev);
      g_wready.erase(old);
    }
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      set_error("hipEventCreate failed");
      return MDCONV_ELAUNCH;
    }
    it = g_wready.emplace(key, WReady{ev, 0}).first;
  }
  it->second.tick = ++g_wready_tick;
  if (hipEventRecord(it->second.ev, stream) != hipSuccess) {
    set_error("hipEventRecord failed");
    return MDCONV_ELAUNCH;
  }
  g_wready_latest[dev] = it->second.ev;
  return MDCONV_OK;
}

static int wait_weight_ready(hipStream_t waiter, bool keyed, hipStream_t producer) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MDCONV_ELAUNCH;
  // the wait is enqueued UNDER the lock: record_weight_ready() may evict and destroy the least recently used event
  // of another stream at any time, and an event copied out of the table could be gone by the time it is waited on
  std::lock_guard<std::mutex> lock(g_wready_mu);
  hipEvent_t ev = nullptr;
  if (keyed) {
    auto it = g_wready.find(std::make_pair(dev, producer));
    if (it != g_wready.end()) ev = it->second.ev;
  } else {
    auto it = g_wready_latest.find(dev);
    if (it != g_wready_latest.end()) ev = it->second;
  }
  if (!ev) {
    set_error(keyed ? "no backward has been issued on that stream of this device"
                    : "no backward has been issued on this device");
    return MDCONV_EINVAL;
  }
  if (hipStreamWaitEvent(waiter, ev, 0) != hipSuccess) {
    set_error("hipStreamWaitEvent failed");
    return MDCONV_ELAUNCH;
  }
  return MDCONV_OK;
}

}  // namespace mdconv

using namespace mdconv;

extern "C" {

int mdconv_abi_version(void) { return MDCONV_ABI_VERSION; }
const char *mdconv_last_error(void) { return g_err; }

int mdconv_out_size(const mdconv_desc *d, int axis) {
  if (!d || axis < 0 || axis > 2) return -1;
  if (axis >= desc_ndim(d)) return 1;
  return (d->in_sz[axis] + 2 * d->pad[axis] - (d->dil[axis] * (d->k_sz[axis] - 1) + 1)) /
             d->stride[axis] + 1;
}

size_t mdconv_workspace_bytes(const mdconv_desc *d, int backward) {
  Geom g;
  Modes md;
  if (fill_geom(d, &g) || call_modes(d, &md)) return 0;
  g.in_cl = md.input_layout == MDCONV_LAYOUT_CHANNELS_LAST ? 1 : 0;   // (the plan of a channels-last call, where the caller says so)
  const bool half = d->dtype == MDCONV_F16 || d->dtype == MDCONV_BF16;
  const size_t direct = backward && half ? direct16_workspace_bytes(g) : 0;   // fp32 copies for the scatter kernels
  if (md.path == MDCONV_PATH_DIRECT) return direct;
  if (hp_supported(g, d->dtype, backward != 0)) {
    const size_t hp = hp_workspace_bytes(g, d->dtype, backward != 0);
    if (backward || hp_forward_preferred(g, d->dtype)) return hp;
    // a few-tile forward: fp32 kernels through fp32 copies, unless the input turns out to be channels-last (not known
    // here): enough for either
    const size_t f32 = mfma_workspace_bytes(g, d->dtype, false);
    return hp > f32 ? hp : f32;
  }
  if (!mfma_supported(g, d->dtype, backward != 0)) return direct;
  return mfma_workspace_bytes(g, d->dtype, backward != 0);
}

int mdconv_set_input_layout(int layout) {
  const int prev = g_input_layout;
  if (layout == MDCONV_LAYOUT_NCHW || layout == MDCONV_LAYOUT_CHANNELS_LAST) g_input_layout = layout;
  return prev;
}

int mdconv_input_layout_supported(const mdconv_desc *d, int layout, int backward) {
  Geom g;
  Modes md;
  if (fill_geom(d, &g) || call_modes(d, &md)) return 0;
  if (layout == MDCONV_LAYOUT_NCHW) return 1;
  if (layout != MDCONV_LAYOUT_CHANNELS_LAST) return 0;
  g.in_cl = 1;   // the plan of a channels-last call (the group-padded layout needs the library's own input copy)
  return md.path != MDCONV_PATH_DIRECT && hp_supported(g, d->dtype, backward != 0) && g.C % 32 == 0;
}

int mdconv_set_accumulate(int on) {
  const int prev = g_accumulate;
  g_accumulate = on ? 1 : 0;
  return prev;
}

int mdconv_stream_wait_weight_ready(void *stream) {
  return wait_weight_ready((hipStream_t)stream, false, nullptr);
}

int mdconv_stream_wait_weight_ready_on(void *stream, void *producer_stream) {
  return wait_weight_ready((hipStream_t)stream, true, (hipStream_t)producer_stream);
}

int mdconv_set_path(int path) {
  const int prev = current_path();
  if (path >= MDCONV_PATH_AUTO && path <= MDCONV_PATH_MFMA) g_path.store(path);
  return prev;
}
int mdconv_last_path(void) { return g_last_path; }
int mdconv_last_kernels(void) { return g_last_kernels; }

int mdconv_deform_conv2d_forward(const mdconv_desc *d, const void *input, const void *weight,
                                 const void *bias, const void *offset, void *output,
                                 void *workspace, size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset; t.output = output;
  return run_forward(d, 2, 0, t, workspace, workspace_bytes, stream);
}

int mdconv_deform_conv2d_backward(const mdconv_desc *d, const void *input, const void *weight,
                                  const void *bias, const void *offset, void *grad_input,
                                  void *grad_weight, void *grad_bias, void *grad_offset,
                                  const void *grad_output, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset;
  t.grad_output = grad_output; t.grad_input = grad_input; t.grad_weight = grad_weight;
  t.grad_bias = grad_bias; t.grad_offset = grad_offset;
  return run_backward(d, 2, 0, t, workspace, workspace_bytes, stream);
}

int mdconv_modulated_deform_conv2d_forward(const mdconv_desc *d, const void *input,
                                           const void *weight, const void *bias,
                                           const void *offset, const void *mask, void *output,
                                           void *workspace, size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset; t.mask = mask;
  t.output = output;
  return run_forward(d, 2, 1, t, workspace, workspace_bytes, stream);
}

int mdconv_modulated_deform_conv2d_backward(const mdconv_desc *d, const void *input,
                                            const void *weight, const void *bias,
                                            const void *offset, const void *mask,
                                            const void *grad_output, void *grad_input,
                                            void *grad_offset, void *grad_mask, void *grad_weight,
                                            void *grad_bias, void *workspace,
                                            size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset; t.mask = mask;
  t.grad_output = grad_output; t.grad_input = grad_input; t.grad_weight = grad_weight;
  t.grad_bias = grad_bias; t.grad_offset = grad_offset; t.grad_mask = grad_mask;
  return run_backward(d, 2, 1, t, workspace, workspace_bytes, stream);
}

int mdconv_deform_conv3d_forward(const mdconv_desc *d, const void *input, const void *weight,
                                 const void *bias, const void *offset, void *output,
                                 void *workspace, size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset; t.output = output;
  return run_forward(d, 3, 0, t, workspace, workspace_bytes, stream);
}

int mdconv_deform_conv3d_backward(const mdconv_desc *d, const void *input, const void *weight,
                                  const void *bias, const void *offset, void *grad_input,
                                  void *grad_weight, void *grad_bias, void *grad_offset,
                                  const void *grad_output, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset;
  t.grad_output = grad_output; t.grad_input = grad_input; t.grad_weight = grad_weight;
  t.grad_bias = grad_bias; t.grad_offset = grad_offset;
  return run_backward(d, 3, 0, t, workspace, workspace_bytes, stream);
}

int mdconv_modulated_deform_conv3d_forward(const mdconv_desc *d, const void *input,
                                           const void *weight, const void *bias,
                                           const void *offset, const void *mask, void *output,
                                           void *workspace, size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset; t.mask = mask;
  t.output = output;
  return run_forward(d, 3, 1, t, workspace, workspace_bytes, stream);
}

int mdconv_modulated_deform_conv3d_backward(const mdconv_desc *d, const void *input,
                                            const void *weight, const void *bias,
                                            const void *offset, const void *mask,
                                            void *grad_input, void *grad_weight, void *grad_bias,
                                            void *grad_offset, void *grad_mask,
                                            const void *grad_output, void *workspace,
                                            size_t workspace_bytes, void *stream) {
  Tensors t = {};
  t.input = input; t.weight = weight; t.bias = bias; t.offset = offset; t.mask = mask;
  t.grad_output = grad_output; t.grad_input = grad_input; t.grad_weight = grad_weight;
  t.grad_bias = grad_bias; t.grad_offset = grad_offset; t.grad_mask = grad_mask;
  return run_backward(d, 3, 1, t, workspace, workspace_bytes, stream);
}

}  // extern "C"
