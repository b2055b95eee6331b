// libganmf_hip.so — handle, step orchestration and the C ABI declared in include/ganmf_hip.h.
//
// One handle = one GPU = one HIP stream (+ one RCCL communicator when data-parallel).  All state
// of a fit() lives in HBM: the CSR user x item matrix, every parameter with its Adam moments and
// best-snapshot twin, the minibatch work buffers.  The host sends one permutation per epoch and
// receives the per-minibatch losses; nothing else crosses PCIe inside the training loop.
//
// Layout in HBM.  Every matrix is row-major with a leading dimension rounded up to 64 floats
// (256-byte rows, whole K-tiles); pad columns of every buffer that is consumed along K are zero
// and stay zero (Adam on a zero gradient of a zero parameter is a fixed point; GEMM / reduce
// epilogues never store into pad columns).
// Bias folding: the encoder bias is stored as row N of We_ext [N+1, e] and the decoder bias as
// row e of Wd_ext [e+1, N]; minibatch matrices carry a ones column (XF[:, N] = 1, E[:, e] = 1).
// The bias add then rides inside the forward GEMMs and both bias gradients fall out of the weight
// gradient GEMMs as their last row, so biases need no kernels of their own (and Adam / L2 see
// them as part of the same tensor, exactly as the reference regularises biases, GANMF.py:131-132).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/ganmf_hip.h"
#include "gemm_f32.hpp"
#include "gemm_bf16s.hpp"
#include "gemm_bf16k.hpp"
#include "gemm_persist.hpp"
#include "gemm_bf16p.hpp"
#include "kernels.hpp"
#include "gemm_multi.hpp"
#ifdef GANMF_PERSIST_DIAG_BUILD
#include "wgrad_stream.hpp"      // experiment (profiles/r04_wgrad_stream.md)
#endif

using namespace ganmf;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(x)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) return fail(-2, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define NCCL_TRY(x)                                                                             \
  do {                                                                                          \
    ncclResult_t r_ = (x);                                                                      \
    if (r_ != ncclSuccess) return fail(-3, "%s failed: %s (%s:%d)", #x, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)
#define TRY(x)            \
  do {                    \
    int rc_ = (x);        \
    if (rc_ != 0) return rc_; \
  } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// K-loop arithmetic: GANMF_MFMA = f32 | bf16x3 | bf16 overrides the handle's choice (tests, A/B timing)
constexpr int MFMA_DEFAULT = MFMA_AUTO;
int env_mfma_mode(int dflt) {
  const char* v = getenv("GANMF_MFMA");
  if (!v || !*v) return dflt;
  if (!strcmp(v, "auto")) return MFMA_AUTO;
  if (!strcmp(v, "f32")) return MFMA_F32;
  if (!strcmp(v, "bf16x3")) return MFMA_BF16X3;
  if (!strcmp(v, "bf16")) return MFMA_BF16;
  if (!strcmp(v, "f16")) return MFMA_F16;
  return dflt;
}

constexpr int ADAM_GRID = 1024;
constexpr int COUNTER_CAP = 1 << 16;

struct Tensor {
  int rows = 0, cols = 0, ld = 0;
  float *p = nullptr, *m = nullptr, *v = nullptr, *best = nullptr, *g = nullptr;
  size_t cap = 0;          // allocated elements: padded() rounded up to world_size equal slices of a multiple of 64 floats
  size_t padded() const { return (size_t)rows * ld; }
  size_t count() const { return (size_t)rows * cols; }
};

// profiling: kernel classes
enum Tag : int {
  T_DENSIFY, T_GEMM_GEN, T_RED_GEN, T_GEMM_ENC, T_RED_ENC, T_GEMM_DEC, T_RED_DEC, T_DCOEF, T_GEMM_DE, T_RED_DE,
  T_GEMM_GWD, T_RED_GWD, T_GEMM_GWE, T_RED_GWE, T_ADAM_D, T_GEMM_DF, T_RED_DF, T_GEMM_GUB, T_RED_GUB, T_GEMM_GV,
  T_RED_GV, T_ADAM_V, T_ADAM_U, T_MULTIRED, T_ALLREDUCE, T_SCORE_GEMM, T_RED_SCORE, T_DIS_FWD, T_RED_DIS_FWD, T_DIS_HEAD,
  T_DIS_GW, T_RED_DIS_GW, T_DIS_BWD, T_RED_DIS_BWD, T_FRONT, T_GWD_RED, T_PAIR, T_DE_DCOEF, T_WPAIR, T_COUNT
};
const char* const kTagName[T_COUNT] = {
  "densify_rows+gather", "gemm_generator[B,k]x[N,k]^T", "reduce_generator", "gemm_encode[2B,N]x[N,e]",
  "reduce_encode", "gemm_decode[2B,e]x[e,N]", "reduce_decode+mse", "d_coef+scale", "gemm_dE[2B,N]x[e,N]^T",
  "reduce_dE", "gemm_gWd[2B,e]^Tx[2B,N]", "reduce_gWd", "gemm_gWe[2B,N]^Tx[2B,e]", "reduce_gWe", "adam_dense_D",
  "gemm_dF[B,e]x[N,e]^T", "reduce_dF", "gemm_gUb[B,N]x[N,k]", "reduce_gUb", "gemm_gV[B,N]^Tx[B,k]", "reduce_gV",
  "adam_dense_V", "adam_rows_U", "multi_reduce", "rccl_allreduce", "gemm_scores", "reduce_scores",
  "gemm_dis_layer_fwd", "reduce_dis_layer_fwd", "dis_head", "gemm_dis_gW", "reduce_dis_gW", "gemm_dis_bwd",
  "reduce_dis_bwd", "gemm_generator[B,k]x[N,k]^T + CSR rows (one launch)", "gemm_gWd[2B,e]^Tx[2B,N] + reduce_dE (one launch)",
  "gemm_gUb[B,N]x[N,k] + gemm_gV[B,N]^Tx[B,k] (one launch)", "gemm_dE[2B,N]x[e,N]^T + d_coef (one launch)",
  "gemm_gWd + gemm_gWe, fused Adam (one launch)"};

struct ProfRec { int tag; hipEvent_t a, b; double flops, bytes; };

}  // namespace

// ---- in-process loopback communicator: group state (see allreduce_local) ----
constexpr int LOCAL_MAX_WORLD = 8;
struct LocalPtrs { float* p[LOCAL_MAX_WORLD]; };
enum LocalOp : int { LOCAL_ALLREDUCE = 0, LOCAL_REDUCE_SCATTER = 1, LOCAL_ALLGATHER = 2 };
// count = elements of the whole buffer (world slices for the scatter / gather forms); sums run in rank order
__global__ void local_collective_kernel(LocalPtrs b, int world, size_t count, int op) {
  const size_t slice = count / (size_t)world;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    if (op == LOCAL_ALLGATHER) {
      const float v = b.p[i / slice][i];           // the owner's copy of its slice
      for (int r = 0; r < world; ++r) b.p[r][i] = v;
      continue;
    }
    float s = b.p[0][i];
    for (int r = 1; r < world; ++r) s += b.p[r][i];
    if (op == LOCAL_ALLREDUCE) { for (int r = 0; r < world; ++r) b.p[r][i] = s; }
    else b.p[i / slice][i] = s;                    // reduce-scatter: only the owner of the slice receives the sum
  }
}
struct LocalGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, dev = -1, arrived = 0, joined = 0;
  long long generation = 0;
  size_t count = 0;
  int op = 0;
  bool failed = false;
  LocalPtrs bufs{};
};

struct ganmf_handle {
  ganmf_cfg cfg;
  int dev = 0;
  hipStream_t st = nullptr;
  hipStream_t st2 = nullptr;             // side lane: independent kernels overlap the main lane
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // ganmf_stream_timer
  hipEvent_t ev_we = nullptr, ev_wd = nullptr;   // data-parallel: "the side lane has finished updating We / Wd" (dp_mark / dp_join)
  int U = 0, N = 0, k = 0, e = 0, B = 0;
  int ldN = 0, ldk = 0, lde = 0;
  // DisGANMF (model 1): hidden layers W_l_ext and the output unit; see the DisGANMF section below
  int L = 0, act = 0;
  std::vector<Tensor> Wl;   // l = 0: [N+2, e] rows 0..N-1 profile weights, row N bias, row N+1 the float(uid) weight; l > 0: [e+1, e]
  Tensor Wo;                // [1, e+1]: output kernel (e) then output bias
  std::vector<float*> Al;   // layer outputs [2B, lde] with the ones column at e
  float *dz0 = nullptr, *dz1 = nullptr, *dlogit = nullptr;
  Tensor We, Wd, Ue, V;   // We = We_ext [N+1, e] (row N = encoder bias), Wd = Wd_ext [e+1, N] (row e = decoder bias)
  float* gD = nullptr;  // contiguous [gWe_ext | gWd_ext] (one all-reduce)
  size_t gD_elems = 0;
  // CSR
  long long* indptr = nullptr;
  int* indices = nullptr;
  float* data = nullptr;
  long long nnz = 0;
  bool has_urm = false;
  bool sparse_g = false;   // generator steps take the real rows' encodings from a CSR row-sum (no densify of X)
  bool sparse_d = false;   //               discriminator steps too: Er from the CSR rows, X subtracted from the reconstruction through a CSR
                           //               lookup in the decode epilogue, X^T.dE_r added to the encoder gradient from the CSC matrix
  long long* csc_colptr = nullptr;   // CSC form of the same matrix (sparse_d): column j = the rows that store item j, ascending
  int* csc_rowidx = nullptr;
  float* csc_val = nullptr;
  float* sp_rows = nullptr;          // [N + CSC_BIAS_PARTS, lde]: X^T . dE_r of the step in flight (csc_rows_kernel)
  // epoch schedule
  int* perm = nullptr;      // [2U] device: the epoch's permutation, then (pos = perm + U) its inverse
  int* pos = nullptr;
  int* stage_i = nullptr;   // pinned host staging (ensure_stage)
  float* stage_f = nullptr;
  size_t stage_i_cap = 0, stage_f_cap = 0;
  // minibatch work buffers
  float *XF = nullptr, *Ub = nullptr, *E = nullptr, *Es = nullptr, *Dl = nullptr, *dE = nullptr, *dF = nullptr, *gUb = nullptr;
  float* zero_page = nullptr;
  float* slab = nullptr;
  size_t slab_elems = 0;
  float* slab2 = nullptr;                // split-K workspace of the side lane
  size_t slab2_elems = 0;
  unsigned *counters = nullptr, *counters2 = nullptr;   // split-K arrival counters (zero between launches)
  int x3kg = 7;                   // GANMF_X3KG bits: 16-wave split-bf16 loop for the plans of the 16-wave fp32 ring kernel (1 NT, 2 K-major B), 4: its one-piece form for bf16 / fp16 plans
  bool inkernel_reduce = true;
  int inlaunch_tags = 0;
  int inlaunch_max = 4;
  float* rs = nullptr;
  float* scal = nullptr;
  float *sqp = nullptr;  // [2][max_tiles]
  int sqp_stride = 0;
  int reg_cap = 0;
  int pair_ring = 2;      // LDS ring depth of the gUb + gV launch (GANMF_PAIR_RING)
  int dis_cap = 0;              // DisGANMF: floats per segment of a step's arena slot (5 + L segments, finish_dis_parts_kernel)
  float* dis_slot = nullptr;    //           slot of the step in flight
  std::vector<char> dis_fused;  // DisGANMF, per layer: this step's update ran in the epilogue of its gradient GEMM (dis_backprop_hidden)
  std::vector<int> dis_regn;    //           and left this many sum(theta^2) partials
  bool defer_gub = true;  // the split-K slabs of gUb are summed by adam_rows_kernel (no reduce launch)
  int multi = 31;         // combined launches (gemm_multi.hpp), bit 0: generator GEMM + CSR row expansion, bit 1: gUb + gV,
                          // bit 2: slab sum of dE inside the gWd launch, bit 3 (with bit 2): d_coef inside the dE launch,
                          // bit 4 (with bit 2): gWd + gWe in one launch behind a stand-alone slab sum of dE (GANMF_MULTI)
  float* V_alt = nullptr; // second parameter buffer of item_embeddings: the fused gV update is written there while gUb
                          // still reads the old V in the same launch; swapped with V.p after the launch
  bool fuse_adam = true;  // single GPU: Adam runs in the epilogue of the weight-gradient GEMMs
  bool dis_fuse_hidden = true;   // DisGANMF: also for the hidden layers l > 0 (GANMF_DIS_FUSE_HIDDEN)
  bool dcoef_spread = true;      // de_dcoef_kernel: d_coef shared out over the GEMM's workgroups instead of extra ones (GANMF_DCOEF_SPREAD)
  float* parts_all = nullptr;                    // ONE allocation: [d_parts | g_parts | d_arena | g_arena], packed per call (parts_begin)
  float *d_parts = nullptr, *g_parts = nullptr;  // [steps][4]
  float* colbuf = nullptr;                       // [cap] one column of d_parts on its way through an all-reduce
  float *d_arena = nullptr, *g_arena = nullptr;  // [cap][4][reg_cap] per-step block partials (GANMF), reduced once per epoch
  int64_t parts_cap = 0;
  // recommend(): URM_train in evaluation orientation for the seen-item mask, top-k outputs
  long long* seen_indptr = nullptr;
  int* seen_indices = nullptr;
  int64_t seen_rows = 0, seen_cols = 0;
  // score filter (ganmf_set_score_filter): byte per score column (1 = computed), and whether cold rows are masked
  unsigned char* item_mask = nullptr;
  size_t item_mask_cap = 0;
  int64_t item_mask_w = 0;             // 0: no item filter; else one past the largest listed item
  bool mask_cold = false;
  int* topk_items = nullptr;
  float* topk_vals = nullptr;
  size_t topk_cap = 0;
  // ganmf_evaluate(): URM_test in evaluation orientation (sorted rows) with the DCG gains, work buffers
  long long* test_indptr = nullptr;
  int* test_indices = nullptr;
  double* test_gain = nullptr;
  int64_t test_rows = 0, test_cols = 0;
  double* eval_buf = nullptr;      // disc | ideal_cum | block partials
  size_t eval_cap = 0;
  // scoring scratch
  int* sc_ids = nullptr;
  size_t sc_ids_cap = 0;
  float *sc_rows = nullptr, *sc_out = nullptr;
  size_t sc_rows_cap = 0, sc_out_cap = 0;
  unsigned *sc_pa = nullptr, *sc_pb = nullptr;      // bf16 x 3 planes of the scored rows / of the other factor (gemm_bf16p.hpp)
  size_t sc_pa_cap = 0, sc_pb_cap = 0;
  const float* sc_pb_src = nullptr;                 // what sc_pb holds: the planes of this parameter buffer ...
  long long sc_pb_version = -1, param_version = 0;  // ... as of this parameter version (bumped by training, set_tensor, restore_best)
  int sc_pb_rows = 0;
  bool score_presplit = true;                       // GANMF_SCORE_PRESPLIT: many-tile scoring products on the pre-split persistent kernel
  // RCCL
  ncclComm_t comm = nullptr;
  bool has_comm = false;
  std::shared_ptr<LocalGroup> local;   // in-process loopback communicator (ganmf_comm_init_local)
  int d_alpha = S_ALPHA_D;             // scalar slot holding lr_t of the discriminator step in flight (alternates in data-parallel runs)
  int side_pending = 0;                // data-parallel: PEND_* bits of the replicated tensors the side lane still updates (dp_join
                                       // before their next use on the main lane)
  bool merge_decode = true;            // GANMF_MERGE_DECODE: the discriminator step's two decode batches as one product (d_step)
  int red_elems = 0;                   // GANMF_RED_ELEMS: float4 outputs per thread of the stand-alone slab sum (0: the fixed 512-block grid)
  // (experiment, make DIAG=1) bf16 x 3 planes of the discriminator step's activations (wgrad_stream.hpp): [2B][ld] per piece, piece stride ps_N / ps_e elements
  bf16raw *pl_XF = nullptr, *pl_Dl = nullptr, *pl_Es = nullptr, *pl_dE = nullptr;
  long long ps_N = 0, ps_e = 0;
  float* wgs_dump = nullptr;           // 8 KiB the stream kernel's out-of-matrix lanes store to
  int* wgs_table = nullptr;            // its tile schedule (wgs_build_schedule), rebuilt when the tile grid changes
  int wgs_key[5] = {0, 0, 0, 0, 0}, wgs_rounds = 0;
  size_t wgs_table_cap = 0;
  bool wgrad_stream = false;           // GANMF_WGRAD_STREAM=1 (make DIAG=1 only): the two fused-Adam weight-gradient products as the persistent role-split launch
  int adam_nfast = 3;                  // GANMF_ADAM_NFAST: tile order of the fused-Adam weight-gradient launch (GemmP::n_fastest)
  long long fork_armed_at = 0;         // launch_count() when fork_arm() handed ev_fork to the next launch
  bool fork_attach = true;             // GANMF_FORK_ATTACH: forks ride on the producing kernel's completion event (fork_arm / fork_wait)
  bool force_coll = false;             // GANMF_FORCE_COLLECTIVES=1: a one-rank communicator still issues its (in-place) reduce-scatter /
                                       // all-gather calls, so that the RCCL call sites execute on a one-GPU box (tests, bench)
  // tuning knobs (environment: GANMF_TILE, GANMF_RING, GANMF_NSPLIT; 0 = cost model decides)
  GemmTune tune;
  bool debug_plan = false;
  int fused_bk = 0;               // K-tile depth of those GEMMs (GANMF_FUSED_BK)
  int fused_tile = 64;            // their output tile (GANMF_FUSED_TILE: 64 | 128)
  int fused_mode = MFMA_BF16X3;   // K-loop arithmetic of the fused-Adam weight-gradient GEMMs under MFMA_AUTO: the two
                                  // [~1000 x ~3700 x 2B] TN GEMMs run 10 % faster on the split-bf16 loop (+2.7 % steps/s)
  std::vector<long long> seen_plans;
  // profiling
  bool prof = false;
  std::vector<ProfRec> recs;
};

namespace {

struct Scope {
  ganmf_handle* h;
  bool on;
  ProfRec r;
  hipStream_t s;
  Scope(ganmf_handle* h_, int tag, double flops, double bytes, hipStream_t st = nullptr)
      : h(h_), on(h_->prof), s(st ? st : h_->st) {
    if (on) {
      r.tag = tag; r.flops = flops; r.bytes = bytes;
      hipEventCreate(&r.a); hipEventCreate(&r.b);
      hipEventRecord(r.a, s);
      launch_prof() = LaunchProf{r.a, r.b, 0};      // the first kernel launched inside the scope stamps both events itself
    }
  }
  ~Scope() {
    if (on) {
      // exactly one kernel: its own start / end are on the events.  None (a collective) or several: bracket as before.
      if (launch_prof().count != 1) hipEventRecord(r.b, s);
      launch_prof() = LaunchProf{};
      h->recs.push_back(r);
    }
  }
};

int dalloc(float** p, size_t elems) {
  HIP_TRY(hipMalloc((void**)p, std::max<size_t>(elems, 4) * sizeof(float)));
  HIP_TRY(hipMemset(*p, 0, std::max<size_t>(elems, 4) * sizeof(float)));
  HIP_TRY(hipDeviceSynchronize());  // the handle's stream does not synchronise with the null stream
  return 0;
}

int alloc_tensor(Tensor& t, int rows, int cols, bool grad_separate, int world, int ld = 0) {
  t.rows = rows; t.cols = cols; t.ld = ld ? ld : round_up(cols + 1, LD_ALIGN);
  // data-parallel runs reduce-scatter the gradient, update one slice per rank and all-gather the parameter: every
  // buffer holds world equal slices (the tail past padded() is zero and stays zero: Adam's fixed point)
  const size_t w = (size_t)std::max(world, 1);
  t.cap = ((t.padded() + w - 1) / w + 63) / 64 * 64 * w;
  TRY(dalloc(&t.p, t.cap));
  TRY(dalloc(&t.m, t.cap));
  TRY(dalloc(&t.v, t.cap));
  TRY(dalloc(&t.best, t.cap));
  if (grad_separate) TRY(dalloc(&t.g, t.cap));
  return 0;
}

void free_tensor(Tensor& t, bool grad_separate) {
  hipFree(t.p); hipFree(t.m); hipFree(t.v); hipFree(t.best);
  if (grad_separate) hipFree(t.g);
}

// A reference variable as up to two row segments of a folded tensor.
struct Seg { Tensor* t; int row0, col0, rows, cols, host_row0; };
struct View { int rows, cols, nseg; Seg seg[2]; };

bool find_view(ganmf_handle* h, int id, View* v) {
  auto one = [&](Tensor* t, int row0, int col0, int rows, int cols) {
    v->rows = rows; v->cols = cols; v->nseg = 1; v->seg[0] = {t, row0, col0, rows, cols, 0};
    return true;
  };
  if (id == GANMF_T_USER_EMB) return one(&h->Ue, 0, 0, h->U, h->k);
  if (id == GANMF_T_ITEM_EMB) return one(&h->V, 0, 0, h->N, h->k);
  if (h->cfg.model == GANMF_MODEL_GANMF) {
    switch (id) {
      case 0: return one(&h->We, 0, 0, h->N, h->e);      // autoencoder/encoding/kernel
      case 1: return one(&h->We, h->N, 0, 1, h->e);      // autoencoder/encoding/bias  (row N of We_ext)
      case 2: return one(&h->Wd, 0, 0, h->e, h->N);      // autoencoder/decoding/kernel
      case 3: return one(&h->Wd, h->e, 0, 1, h->N);      // autoencoder/decoding/bias  (row e of Wd_ext)
      default: return false;
    }
  }
  // DisGANMF: 2l layer_l/kernel, 2l+1 layer_l/bias, 2L D_output/kernel [e,1], 2L+1 D_output/bias [1]
  if (id < 0 || id > 2 * h->L + 1) return false;
  if (id == 2 * h->L) { v->rows = h->e; v->cols = 1; v->nseg = 1; v->seg[0] = {&h->Wo, 0, 0, 1, h->e, 0}; return true; }
  if (id == 2 * h->L + 1) return one(&h->Wo, 0, h->e, 1, 1);
  const int l = id / 2;
  Tensor* t = &h->Wl[l];
  if (id & 1) return one(t, l == 0 ? h->N : h->e, 0, 1, h->e);
  if (l > 0) return one(t, 0, 0, h->e, h->e);
  // layer_0/kernel is [N+1, e] in the reference with row 0 multiplying float(uid) (DisGANMF.py:59)
  v->rows = h->N + 1; v->cols = h->e; v->nseg = 2;
  v->seg[0] = {t, h->N + 1, 0, 1, h->e, 0};
  v->seg[1] = {t, 0, 0, h->N, h->e, 1};
  return true;
}

std::vector<Tensor*> all_tensors(ganmf_handle* h) {
  std::vector<Tensor*> v = {&h->Ue, &h->V};
  if (h->cfg.model == GANMF_MODEL_GANMF) { v.push_back(&h->We); v.push_back(&h->Wd); }
  else { for (auto& t : h->Wl) v.push_back(&t); v.push_back(&h->Wo); }
  return v;
}

float* slot_ptr(Tensor* t, int slot) {
  switch (slot) {
    case GANMF_SLOT_PARAM: return t->p;
    case GANMF_SLOT_ADAM_M: return t->m;
    case GANMF_SLOT_ADAM_V: return t->v;
    case GANMF_SLOT_BEST: return t->best;
    default: return nullptr;
  }
}

// MFMA_F16: power of two that brings an operand carrying the loss gradient's 1/(B.N) (GANMF: mean over B.N reconstruction
// errors) or 1/B (DisGANMF: mean over B cross-entropies, times an output weight of ~2^-5) into fp16's normal range
inline float grad_scale(const ganmf_handle* h, int b_global) {
  if (h->tune.mode != MFMA_F16) return 0.f;
  const float lg = h->cfg.model == GANMF_MODEL_GANMF ? log2f((float)b_global * (float)h->N) : log2f((float)b_global) + 6.f;
  return exp2f(roundf(lg));
}
inline bool low_precision(const ganmf_handle* h) { return h->tune.mode == MFMA_F16 || h->tune.mode == MFMA_BF16; }

inline double gemm_flops(double M, double N, double K) { return 2.0 * M * N * K; }
inline double gemm_bytes(double M, double N, double K) { return 4.0 * (M * K + N * K + M * N); }

// ---- in-process loopback communicator ---------------------------------------------------------------
// Several handles of ONE process (driven by one host thread each) on the same device form a group; an all-reduce
// is a rendezvous: every member drains the stream its buffer is produced on, the last to arrive sums the buffers in
// rank order (deterministic) into every member's buffer and releases the others.  It exists so that the
// data-parallel arithmetic (global-batch scales, the two-float exchange before the hinge, ranks that run out of
// rows) can be exercised with world_size > 1 on a single GPU; multi-GPU runs use RCCL.
static std::mutex g_local_mu;
static std::map<int, std::shared_ptr<LocalGroup>> g_local_groups;

int collective_local(ganmf_handle* h, float* buf, size_t count, hipStream_t st, int op) {
  LocalGroup& g = *h->local;
  HIP_TRY(hipStreamSynchronize(st));                    // this member's contribution is complete
  std::unique_lock<std::mutex> lk(g.mu);
  if (g.failed) return fail(-3, "local communicator: a peer failed");
  if (g.arrived == 0) { g.count = count; g.op = op; }
  else if (g.count != count || g.op != op) { g.failed = true; g.cv.notify_all(); return fail(-3, "local communicator: members disagree on the collective"); }
  g.bufs.p[h->cfg.rank] = buf;
  if (++g.arrived == g.world) {
    const int grid = (int)std::min<size_t>(1024, (count + 255) / 256);
    GANMF_LAUNCH(local_collective_kernel, dim3(std::max(grid, 1)), dim3(256), 0, st, g.bufs, g.world, count, op);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    g.arrived = 0;
    ++g.generation;
    if (e != hipSuccess) g.failed = true;
    g.cv.notify_all();
    if (e != hipSuccess) return fail(-2, "local communicator: %s", hipGetErrorString(e));
    return 0;
  }
  const long long gen = g.generation;
  if (!g.cv.wait_for(lk, std::chrono::seconds(120), [&] { return g.generation != gen || g.failed; })) {
    g.failed = true;
    g.cv.notify_all();
    return fail(-3, "local communicator: timed out waiting for %d peer(s)", g.world - g.arrived);
  }
  return g.failed ? fail(-3, "local communicator: a peer failed") : 0;
}

int allreduce(ganmf_handle* h, float* buf, size_t count, int lane = 0) {
  if (!h->has_comm) return 0;
  hipStream_t st = lane ? h->st2 : h->st;
  Scope s(h, T_ALLREDUCE, 0, 4.0 * count, st);
  if (h->local) return collective_local(h, buf, count, st, LOCAL_ALLREDUCE);
  NCCL_TRY(ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, h->comm, st));
  return 0;
}

// In place over a buffer of world equal slices: after the reduce-scatter rank r holds the sum of slice r (the other
// slices of its buffer are unspecified); the all-gather fills every slice from its owner.
int reduce_scatter(ganmf_handle* h, float* buf, size_t total, int lane) {
  if (h->cfg.world_size == 1 && !h->force_coll) return 0;     // one slice, already in place
  hipStream_t st = lane ? h->st2 : h->st;
  const size_t slice = total / (size_t)h->cfg.world_size;
  Scope s(h, T_ALLREDUCE, 0, 4.0 * total, st);
  if (h->local) return collective_local(h, buf, total, st, LOCAL_REDUCE_SCATTER);
  NCCL_TRY(ncclReduceScatter(buf, buf + (size_t)h->cfg.rank * slice, slice, ncclFloat, ncclSum, h->comm, st));
  return 0;
}
int all_gather(ganmf_handle* h, float* buf, size_t total, int lane) {
  if (h->cfg.world_size == 1 && !h->force_coll) return 0;
  hipStream_t st = lane ? h->st2 : h->st;
  const size_t slice = total / (size_t)h->cfg.world_size;
  Scope s(h, T_ALLREDUCE, 0, 4.0 * total, st);
  if (h->local) return collective_local(h, buf, total, st, LOCAL_ALLGATHER);
  NCCL_TRY(ncclAllGather(buf + (size_t)h->cfg.rank * slice, buf, slice, ncclFloat, h->comm, st));
  return 0;
}

// Blocks of a stand-alone slab sum over an [M, N] output: one float4 of output per thread (GANMF_RED_ELEMS per thread), at most
// GEMM_RED_GRID.  The step's outputs are 32 k - 118 k float4: a fixed 512-block grid left up to 3/4 of its threads without work
// and the launch paid their dispatch.  (Also the number of sum-of-squares partials such a launch writes.)
int red_grid(const ganmf_handle* h, long long M, long long N, int nsplit) {
  const long long total4 = M * ((N + 3) / 4);
  // (a deep split behind a small output is summed by reduce_groups() threads per element: 256 / G elements per block and pass)
  const long long per_block = 256LL / reduce_groups(total4, nsplit) * std::max(1, h->red_elems);
  return (int)std::max<long long>(1, std::min<long long>(GEMM_RED_GRID, (total4 + per_block - 1) / per_block));
}

int ensure_slab(ganmf_handle* h, size_t elems, int lane) {
  float*& slab = lane ? h->slab2 : h->slab;
  size_t& cap = lane ? h->slab2_elems : h->slab_elems;
  if (elems <= cap) return 0;
  HIP_TRY(hipStreamSynchronize(h->st));
  HIP_TRY(hipStreamSynchronize(h->st2));
  if (slab) hipFree(slab);
  slab = nullptr; cap = 0;
  const size_t want = elems + elems / 4 + 1024;
  TRY(dalloc(&slab, want));
  cap = want;
  return 0;
}

// One logical GEMM of the step: plan (tile / ring / split-K), launch, and when split the reduce
// kernel that applies the epilogue.  *sq_count = partial sums per batch written to epi.sq_partials.
// `defer` (plain-store GEMMs only): a split product is left as its slabs for the CONSUMER kernel to sum in split order
// (one launch less); the slabs live in the second workspace so that the GEMMs in between may use the first.
struct SlabRef { const float* p; int nsplit; long long split_stride; };

// `defer_red`: a split product is launched WITHOUT its reduce kernel; the RedP that finishes it (slab sum + epilogue) is
// handed back (part != nullptr) for the caller to attach to a later launch.  `attach`: such a RedP of ANOTHER product, run
// as extra blocks of this GEMM's launch when the plan allows (gemm_bf16s_red), else as its own kernel first.
int run_gemm(ganmf_handle* h, int tag_gemm, int tag_red, GemmP g, bool akm, bool bkm, int* sq_count = nullptr,
             double extra_bytes = 0, int lane = 0, const GemmTune* force = nullptr, SlabRef* defer = nullptr,
             RedP* defer_red = nullptr, const RedP* attach = nullptr) {
  if (g.nbatch < 1) g.nbatch = 1;
  g.zero_page = h->zero_page;
  if (g.epi.kind == EPI_ADAM && (h->adam_nfast & 2)) g.n_fastest = 1;      // (GANMF_ADAM_NFAST bit 1: every fused-Adam product, not only the paired launch)
  hipStream_t st = lane ? h->st2 : h->st;
  const GemmTune& tn = force ? *force : h->tune;
  GemmPlan pl = gemm_plan(g.M, g.N, g.K, g.nbatch, g.epi.sq_partials != nullptr, tn, g.epi.kind == EPI_ADAM);
  pl.persist = gemm_persist_eligible(g, akm, bkm, pl, tn.persist) ? (tn.persist >= 2 ? tn.persist : 1) : 0;
  // a plan for the 16-wave fp32 ring kernel (one 64 x 64 tile per CU) runs the 16-wave split-bf16 loop instead: same grid, same split,
  // same epilogue, 6 / 16 of the MFMA cycles (gemm_bf16k.hpp)
  // (only under MFMA_AUTO: a handle created with GANMF_FLAG_MFMA_F32 / mfma="f32" runs the fp32 MFMA everywhere, as documented)
  if (!force && h->tune.mode == MFMA_AUTO && pl.mode == MFMA_F32 && pl.tile == 64 && pl.kg == 4 && pl.ring == 3 && !pl.persist && !(akm && !bkm) &&
      tag_gemm != T_GEMM_GUB && tag_gemm != T_GEMM_GV &&      // (gUb / gV: their other form is the fp32 pair_kernel)
      (((h->x3kg & 1) && !akm && !bkm) || ((h->x3kg & 2) && bkm))) pl.mode = MFMA_BF16X3;      // bit 0: NT products, bit 1: products with a K-major B
  // ... and a single-piece (bf16 / fp16) plan with a CU per workgroup takes the same 16-wave kernel in its one-piece form (bit 2)
  if (!force && (h->x3kg & 4) && (pl.mode == MFMA_F16 || pl.mode == MFMA_BF16) && pl.tile == 64 && pl.ring == 3 && pl.bk != 32 && !pl.persist &&
      !(akm && !bkm)) pl.kg = 4;
  if (defer) *defer = SlabRef{g.C, 1, 0};
  // (a deep split behind a small output is summed 16 threads per element by the reduce kernel: not left to the consumer)
  const bool deferred = defer && pl.nsplit > 1 && g.epi.kind == EPI_STORE && g.nbatch == 1 &&
                        reduce_groups((long long)g.M * ((g.N + 3) >> 2), pl.nsplit) == 1;
  const int slab_lane = deferred ? 1 : lane;
  if (pl.nsplit > 1) TRY(ensure_slab(h, gemm_slab_elems(pl, g.M, g.ldc, g.nbatch), slab_lane));
  float* slab = slab_lane ? h->slab2 : h->slab;
  const size_t slab_elems = slab_lane ? h->slab2_elems : h->slab_elems;
  // in-launch split-K reduction (the last workgroup to arrive at a tile sums its slabs, gemm_f32.hpp): for every product
  // (GANMF_INKERNEL_REDUCE=1) or for the classes of GANMF_INLAUNCH_TAGS (1 encode, 2 decode, 4 dF, 8 dE of the generator step)
  const int tag_bit = tag_gemm == T_GEMM_ENC ? 1 : tag_gemm == T_GEMM_DEC ? 2 : tag_gemm == T_GEMM_DF ? 4 : tag_gemm == T_GEMM_DE ? 8 : 0;
  unsigned* counters = (h->inkernel_reduce || (h->inlaunch_tags & tag_bit)) ? (lane ? h->counters2 : h->counters) : nullptr;
  const size_t n_tiles = (size_t)pl.tiles_m * pl.tiles_n * g.nbatch;
  // the last-arriving workgroup reads nsplit slab tiles alone (~60-120 GB/s per block): in-launch
  // reduction only pays for shallow splits; deep splits keep the chip-wide reduce kernel
  const bool in_launch = !deferred && pl.nsplit > 1 && pl.nsplit <= h->inlaunch_max && counters && n_tiles <= (size_t)COUNTER_CAP;
  const bool wants_sq = g.epi.sq_partials != nullptr;
  const int nredg = h->red_elems > 0 ? red_grid(h, g.M, g.N, pl.nsplit) : GEMM_RED_GRID;      // blocks (and partials) of a stand-alone slab sum
  const int sqc = !wants_sq ? 0 : (pl.nsplit > 1 && !in_launch ? nredg : pl.sq_count);
  if (sq_count) *sq_count = sqc;
  if (h->debug_plan) {
    const long long key = ((long long)tag_gemm << 48) ^ ((long long)g.M << 32) ^ ((long long)g.N << 16) ^ g.K;
    if (std::find(h->seen_plans.begin(), h->seen_plans.end(), key) == h->seen_plans.end()) {
      h->seen_plans.push_back(key);
      fprintf(stderr, "[ganmf plan] %-28s M=%d N=%d K=%d batch=%d -> tile %d ring %d kg %d nsplit %d (kps %d) mfma %s wgs %d est %.1f us%s\n",
              kTagName[tag_gemm], g.M, g.N, g.K, g.nbatch, pl.tile, pl.ring, pl.kg, pl.nsplit, pl.kps,
              pl.mode == MFMA_BF16X3 ? "bf16x3" : pl.mode == MFMA_BF16 ? "bf16" : pl.mode == MFMA_F16 ? "f16" : "f32",
              pl.tiles_m * pl.tiles_n * pl.nsplit * g.nbatch, pl.est_us,
              pl.persist ? " (persistent tile walk)" : in_launch ? " (in-launch reduce)" : "");
    }
  }
  const double fl = g.nbatch * gemm_flops(g.M, g.N, g.K), by = gemm_bytes((double)g.nbatch * g.M, g.N, g.K) + extra_bytes;
  if (defer_red) defer_red->part = nullptr;
  if (attach && attach->part) {
    const int n4 = (attach->N + 3) / 4;
    const int nred = (int)std::min<long long>(GEMM_RED_GRID, ((long long)attach->M * n4 + 255) / 256);
    const bool combined = pl.mode == MFMA_BF16X3 && pl.tile == 64 && pl.bk == 32 && pl.nsplit == 1 && akm && bkm &&
                          !pl.persist && g.nbatch == 1 && attach->epi.sq_partials == nullptr;
    if (combined) {
      Scope s(h, T_GWD_RED, fl, by + 4.0 * (attach->nsplit + 1) * attach->M * attach->N, st);
      GemmP q = g;
      fill_plan(q, pl);
      const int ng = pl.tiles_m * pl.tiles_n;
      GANMF_LAUNCH(gemm_bf16s_red, dim3(ng + nred), dim3(256), 0, st, q, *attach, nred);
      HIP_TRY(hipGetLastError());
      return 0;
    }
    {
      Scope s(h, tag_red, 0, 4.0 * (attach->nsplit + 1) * attach->M * attach->N, st);
      RedP r = *attach;
      GANMF_LAUNCH(splitk_reduce_kernel, dim3(h->red_elems > 0 ? red_grid(h, r.M, r.N, r.nsplit) : GEMM_RED_GRID, 1), dim3(256), 0, st, r);
      HIP_TRY(hipGetLastError());
    }
  }
  if (defer_red && pl.nsplit > 1 && !in_launch && !deferred && g.nbatch == 1) {    // GEMM now, its reduce rides in a later launch
    Scope s(h, tag_gemm, fl, by, st);
    GemmP q = g;
    fill_plan(q, pl);
    q.C = slab; q.c_split_stride = (long long)g.M * g.ldc; q.c_batch_stride = (long long)g.M * g.ldc;
    HIP_TRY(gemm_dispatch(st, q, akm, bkm, pl));
    RedP r{};
    r.part = slab; r.nsplit = pl.nsplit; r.out = g.C; r.ld = g.ldc; r.M = g.M; r.N = g.N;
    r.batch_stride = g.c_batch_stride; r.epi = g.epi; r.epi.sq_stride = nredg;
    r.split_stride = (long long)g.M * g.ldc;
    *defer_red = r;
    if (sq_count && wants_sq) *sq_count = nredg;
    return 0;
  }
  if (deferred) {     // slabs only: the consumer sums them
    Scope s(h, tag_gemm, fl, by + 4.0 * pl.nsplit * g.M * g.N, st);
    GemmP q = g;
    q.tiles_m = pl.tiles_m; q.tiles_n = pl.tiles_n; q.nsplit = pl.nsplit; q.k_per_split = pl.kps; q.counters = nullptr;
    q.C = slab; q.c_split_stride = (long long)g.M * g.ldc; q.c_batch_stride = (long long)g.M * g.ldc;
    HIP_TRY(gemm_dispatch(st, q, akm, bkm, pl));
    *defer = SlabRef{slab, pl.nsplit, (long long)g.M * g.ldc};
    return 0;
  }
  if (!h->prof || pl.nsplit == 1 || in_launch) {
    Scope s(h, tag_gemm, fl, by + (pl.nsplit > 1 ? 8.0 * pl.nsplit * g.nbatch * g.M * g.N : 0), st);
    HIP_TRY(gemm_run(st, g, akm, bkm, pl, slab, slab_elems, in_launch ? counters : nullptr, COUNTER_CAP, nredg));
    return 0;
  }
  {  // profiled, separate reduce kernel: bracket the GEMM and the reduce separately
    Scope s(h, tag_gemm, fl, by, st);
    GemmP q = g;
    q.tiles_m = pl.tiles_m; q.tiles_n = pl.tiles_n; q.nsplit = pl.nsplit; q.k_per_split = pl.kps; q.counters = nullptr;
    q.C = slab; q.c_split_stride = (long long)g.nbatch * g.M * g.ldc; q.c_batch_stride = (long long)g.M * g.ldc;
    hipError_t e;
    e = gemm_dispatch(st, q, akm, bkm, pl);
    HIP_TRY(e);
  }
  {
    Scope s(h, tag_red, 0, 4.0 * (pl.nsplit + 1) * g.nbatch * g.M * g.N, st);
    RedP r{};
    r.part = slab; r.nsplit = pl.nsplit; r.out = g.C; r.ld = g.ldc; r.M = g.M; r.N = g.N;
    r.batch_stride = g.c_batch_stride; r.epi = g.epi; r.epi.sq_stride = nredg;
    r.split_stride = (long long)g.nbatch * g.M * g.ldc;
    GANMF_LAUNCH(splitk_reduce_kernel, dim3(nredg, g.nbatch), dim3(256), 0, st, r);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// the combined launches (front_kernel, de_dcoef_kernel) carry the 16-wave split-bf16 loop where the stand-alone product would run it:
// GANMF_X3KG bit 0, and only when the handle leaves the arithmetic to the planner (a forced fp32-MFMA handle keeps the fp32 MFMA)
inline bool combined_x3(const ganmf_handle* h) { return (h->x3kg & 1) != 0 && h->tune.mode == MFMA_AUTO; }

// fork: the side lane starts after everything enqueued so far on the main lane
int lane_fork(ganmf_handle* h) {
  HIP_TRY(hipEventRecord(h->ev_fork, h->st));
  HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
  return 0;
}
// The same fork in two halves around the LAST main-lane launch the side lane has to wait for: fork_arm() before it (that launch
// then carries ev_fork as its completion event: no marker packet on the main lane), fork_wait() after it.  While the library's
// profiler is on (its scopes hand their own events to the launches) and with GANMF_FORK_ATTACH=0 the plain fork is used.
// The attached event marks the FIRST launch after fork_arm(): the fork is only taken as attached when exactly one kernel was
// launched in between (a plan that adds a stand-alone reduce or an attached slab sum falls back to the marker-packet fork, which
// covers everything enqueued so far -- never a silent wait for the wrong kernel).
bool fork_arm(ganmf_handle* h) {
  if (!h->fork_attach || h->prof) return false;
  launch_stop_event() = h->ev_fork;
  h->fork_armed_at = launch_count();
  return true;
}
int fork_wait(ganmf_handle* h, bool armed) {
  if (!armed) return lane_fork(h);
  const long long launched = launch_count() - h->fork_armed_at;
  if (launch_stop_event() != nullptr || launched != 1) {      // nothing, or more than one kernel, was launched in between: plain fork
    launch_stop_event() = nullptr;
    return lane_fork(h);
  }
  HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
  return 0;
}
// join: the main lane continues after everything enqueued so far on the side lane
int lane_join(ganmf_handle* h) {
  HIP_TRY(hipEventRecord(h->ev_join, h->st2));
  HIP_TRY(hipStreamWaitEvent(h->st, h->ev_join, 0));
  return 0;
}

// The main lane may not touch We / Wd again before the side lane's reduce-scatter / Adam / all-gather on them are done.
// dp_mark: everything enqueued on the side lane so far completes the update of the tensors in `mask`; dp_join: the main lane
// waits for the tensors in `mask` that are still pending -- the encoder before the next encode GEMM, the decoder only before the
// next decode GEMM, one encode GEMM later (the side lane is in order: the decoder's event implies the encoder's).
enum : int { PEND_WE = 1, PEND_WD = 2, PEND_ALL = 3 };
int dp_mark(ganmf_handle* h, int mask) {
  if (mask & PEND_WE) HIP_TRY(hipEventRecord(h->ev_we, h->st2));
  if (mask & PEND_WD) HIP_TRY(hipEventRecord(h->ev_wd, h->st2));
  h->side_pending |= mask;
  return 0;
}
int dp_join(ganmf_handle* h, int mask = PEND_ALL) {
  const int m = h->side_pending & mask;
  if (!m) return 0;
  if (m & PEND_WE) HIP_TRY(hipStreamWaitEvent(h->st, h->ev_we, 0));
  if (m & PEND_WD) HIP_TRY(hipStreamWaitEvent(h->st, h->ev_wd, 0));
  h->side_pending &= ~m;
  return 0;
}

int adam_dense(ganmf_handle* h, int tag, Tensor& t, const float* g, int alpha_idx, float reg, float* sq, int lane = 0,
               size_t off = 0, size_t count = 0) {
  if (count == 0) count = t.padded() - off;
  const long long n4 = (long long)count / 4;
  hipStream_t st = lane ? h->st2 : h->st;
  Scope s(h, tag, 0, 28.0 * count, st);
  GANMF_LAUNCH(adam_dense_kernel, dim3(ADAM_GRID), dim3(256), 0, st, t.p + off, t.m + off, t.v + off, g + off, n4, h->scal,
                     alpha_idx, reg, sq);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Data-parallel update of a replicated tensor whose local gradient is complete in t.g: reduce-scatter the
// gradient, TF-Adam on this rank's slice only (parameter and moment slices; the moments of a replicated tensor live
// sharded across the ranks), all-gather the parameter.  Same bytes on the links as an all-reduce, 1 / world of the Adam
// traffic, and every rank ends with bitwise identical parameters (each slice has exactly one writer).  The sum(theta^2)
// partials cover the slice only: the loss parts are all-reduced once per epoch.
int dp_update(ganmf_handle* h, int tag, Tensor& t, int alpha_idx, float reg, float* sq, int lane) {
  const size_t slice = t.cap / (size_t)h->cfg.world_size, off = (size_t)h->cfg.rank * slice;
  TRY(reduce_scatter(h, t.g, t.cap, lane));
  TRY(adam_dense(h, tag, t, t.g, alpha_idx, reg, sq, lane, off, slice));
  TRY(all_gather(h, t.p, t.cap, lane));
  return 0;
}

// CSR row expansion of the real rows (+ ones column, + DisGANMF's float(uid) column), embedding lookup Ub = U[uids] and the
// generator GEMM F = Ub . V^T -> rows [nb, 2nb) of XF (GANMF.py:82-83,183-184).  One launch when the generator GEMM is
// planned onto the 16-wave fp32 ring kernel (front_kernel: the GEMM fetches its A rows from U through the row list, the
// row expansion runs as extra workgroups of the same grid); else the two kernels one after the other.
int rows_and_generator(ganmf_handle* h, const int* rows_dev, int nb, int which, int aslot, int uid_col, int row_offset) {
  const int N = h->N, k = h->k;
  DensP d{h->indptr, h->indices, h->data, rows_dev, nb, N, h->XF, h->ldN, h->Ue.p, h->ldk, h->Ub, h->scal, which, aslot,
          which ? h->cfg.g_lr : h->cfg.d_lr, uid_col, row_offset};
  GemmP g{};
  g.A = h->Ub; g.lda = h->ldk; g.B = h->V.p; g.ldb = h->ldk;
  g.C = h->XF + (size_t)nb * h->ldN; g.ldc = h->ldN;
  g.M = nb; g.N = N; g.K = k; g.epi.kind = EPI_STORE; g.nbatch = 1;
  g.zero_page = h->zero_page;
  // K = num_factors is at most four K-tiles: a split would trade three of them for a reduce launch and keep the GEMM out of
  // the combined launch (B = 64 at ML-1M shape: 4.5 + 5.2 + 4.6 us as three kernels, ~8 us as one)
  GemmPlan pl = gemm_plan(g.M, g.N, g.K, 1, false, h->tune, /*no_split=*/(h->multi & 1) != 0 && k <= 256);
  pl.persist = gemm_persist_eligible(g, false, false, pl, h->tune.persist) ? 1 : 0;
  if ((h->multi & 1) && plan_is_f32_64_kg(pl, 4) && pl.nsplit == 1) {
    if (h->debug_plan) {
      const long long key = ((long long)T_GEMM_GEN << 48) ^ ((long long)g.M << 32) ^ ((long long)g.N << 16) ^ g.K;
      if (std::find(h->seen_plans.begin(), h->seen_plans.end(), key) == h->seen_plans.end()) {
        h->seen_plans.push_back(key);
        fprintf(stderr, "[ganmf plan] %-28s M=%d N=%d K=%d batch=1 -> tile 64 ring 3 kg 4 nsplit 1 (kps %d) mfma %s wgs %d est %.1f us (one launch with the %d CSR rows)\n",
                kTagName[T_GEMM_GEN], g.M, g.N, g.K, pl.kps, combined_x3(h) ? "bf16x3" : "f32", pl.tiles_m * pl.tiles_n, pl.est_us, nb);
      }
    }
    Scope s(h, T_FRONT, gemm_flops(g.M, g.N, g.K), gemm_bytes(g.M, g.N, g.K) + 4.0 * nb * (N + 2 * k));
    g.A = h->Ue.p; g.a_gather = rows_dev;        // A row r = U[rows[r]]: the lookup rides in the operand fetch
    fill_plan(g, pl);
    if (combined_x3(h)) GANMF_LAUNCH((front_kernel<4, true>), dim3(pl.tiles_m * pl.tiles_n + nb), dim3(1024), 0, h->st, g, d);
    else GANMF_LAUNCH(front_kernel<4>, dim3(pl.tiles_m * pl.tiles_n + nb), dim3(1024), 0, h->st, g, d);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  {
    Scope s(h, T_DENSIFY, 0, 4.0 * nb * (N + 2 * k));
    d.nseg = std::max(1, std::min(16, (h->ldN + 8191) / 8192));      // wide rows (C4: 50 000 columns): several workgroups per row
    GANMF_LAUNCH(densify_rows_kernel, dim3(nb * d.nseg), dim3(256), 0, h->st, d);
    HIP_TRY(hipGetLastError());
  }
  return run_gemm(h, T_GEMM_GEN, T_RED_GEN, g, false, false);
}

// ---- shared front of both steps: X rows (+ones column), Ub, F, E = [X;F|1].We_ext ----------------
int step_front(ganmf_handle* h, const int* rows_dev, int nb, int which) {
  const int N = h->N, k = h->k, e = h->e;
  const int aslot = which ? S_ALPHA_G : h->d_alpha;
  if ((which == 1 && h->sparse_g) || (which == 0 && h->sparse_d)) {
    // sparse regime: Er from the CSR rows, X never materialised, the encode GEMM runs on the generated half only
    TRY(dp_join(h, PEND_WE));
    {
      Scope s(h, T_DENSIFY, 0, 4.0 * nb * (2 * k + e) + 4.0 * (double)h->nnz / std::max(h->U, 1) * nb * e);
      GANMF_LAUNCH(sparse_front_kernel, dim3(nb), dim3(256), 0, h->st, h->indptr, h->indices, h->data, rows_dev, nb, N,
                         h->XF, h->ldN, h->Ue.p, h->ldk, h->Ub, h->scal, which, aslot, which ? h->cfg.g_lr : h->cfg.d_lr, h->We.p, h->lde, e, h->E);
      HIP_TRY(hipGetLastError());
    }
    {
      GemmP g{};
      g.A = h->Ub; g.lda = h->ldk; g.B = h->V.p; g.ldb = h->ldk;
      g.C = h->XF + (size_t)nb * h->ldN; g.ldc = h->ldN;
      g.M = nb; g.N = N; g.K = k; g.epi.kind = EPI_STORE;
      TRY(run_gemm(h, T_GEMM_GEN, T_RED_GEN, g, false, false));
    }
    {
      TRY(dp_join(h, PEND_WE));
      GemmP g{};
      g.A = h->XF + (size_t)nb * h->ldN; g.lda = h->ldN; g.B = h->We.p; g.ldb = h->lde;
      g.C = h->E + (size_t)nb * h->lde; g.ldc = h->lde; g.M = nb; g.N = e; g.K = N + 1; g.epi.kind = EPI_STORE;
      TRY(run_gemm(h, T_GEMM_ENC, T_RED_ENC, g, false, true));
    }
    return 0;
  }
  TRY(rows_and_generator(h, rows_dev, nb, which, aslot, -1, 0));   // X rows, Ub, F = Ub . V^T -> rows [nb, 2nb) of XF  (GANMF.py:82-83)
  {  // E = [X;F | 1] . We_ext  (bias = row N); the ones column E[:, e] is never overwritten  (GANMF.py:64-65)
    TRY(dp_join(h, PEND_WE));     // (data-parallel: the previous step's encoder update ran on the side lane under densify + generator GEMM)
    GemmP g{};
    g.A = h->XF; g.lda = h->ldN; g.B = h->We.p; g.ldb = h->lde;
    g.C = h->E; g.ldc = h->lde; g.M = 2 * nb; g.N = e; g.K = N + 1; g.epi.kind = EPI_STORE;
    TRY(run_gemm(h, T_GEMM_ENC, T_RED_ENC, g, false, true));
  }
  return 0;
}

// One discriminator update on local rows rows_dev[0..nb) (GANMF.py:131-132,138,186-187).
int d_step(ganmf_handle* h, const int* rows_dev, int nb, int b_global, float* parts, float* arena) {
  float* regWe = arena + 2 * (size_t)h->reg_cap;   // this step's arena slot: seg2 = sum We_ext^2, seg3 = sum Wd_ext^2
  float* regWd = arena + 3 * (size_t)h->reg_cap;
  const int N = h->N, e = h->e;
  const bool dist = h->has_comm;
  if (dist) h->d_alpha = h->d_alpha == S_ALPHA_D ? S_ALPHA_D_ALT : S_ALPHA_D;
  const int aslot = h->d_alpha;
  const float inv_bn = 1.0f / ((float)b_global * (float)N);
  int sqn = 0;
  bool fused = false;
  int regn[2] = {ADAM_GRID, ADAM_GRID};
  if (nb > 0) {
    TRY(step_front(h, rows_dev, nb, 0));
    TRY(dp_join(h, PEND_WD));      // (data-parallel: the previous step's decoder update ran under this step's front and encode GEMM)
    {  // Delta = [E|1].Wd_ext - inp, per-path sum of squares  (GANMF.py:66-68), batch z = path
      GemmP g{};
      g.A = h->E; g.lda = h->lde; g.B = h->Wd.p; g.ldb = h->ldN;
      g.C = h->Dl; g.ldc = h->ldN; g.M = nb; g.N = N; g.K = e + 1;
      g.nbatch = 2; g.a_batch_stride = (long long)nb * h->lde; g.c_batch_stride = (long long)nb * h->ldN;
      g.epi.kind = EPI_SUB_AUX_SQ; g.epi.aux = h->XF; g.epi.ldaux = h->ldN;
      g.epi.aux_batch_stride = (long long)nb * h->ldN; g.epi.sq_partials = h->sqp;
      if (h->sparse_d) {      // the real rows were never expanded: X[m, n] is looked up in the CSR rows of the batch
        g.epi.csr_indptr = h->indptr; g.epi.csr_indices = h->indices; g.epi.csr_data = h->data; g.epi.csr_rows = rows_dev;
      }
      // [real ; generated] are contiguous in E, XF and Dl: with whole 64-row tiles per path the two batches are ONE product of
      // 2.nb rows, whose list order (tile row fastest) puts the four row tiles of a tile column next to each other on one XCD --
      // the decoder panel is fetched once instead of once per batch (rocprofv3 FETCH_SIZE: 42.5 MB per launch as two batches
      // against 23.3 MB algorithmic, profiles/r02_traffic.json).  Same tiles, same K order, partial sums filed per path as before.
      bool merged = false;
      if (!h->sparse_d && nb % 64 == 0 && h->merge_decode) {
        const GemmPlan p1 = gemm_plan(2 * nb, g.N, g.K, 1, true, h->tune), p2 = gemm_plan(g.M, g.N, g.K, 2, true, h->tune);
        if (p1.tile == 64 && p2.tile == 64 && p1.nsplit == 1 && p2.nsplit == 1 && p1.mode == p2.mode && p1.kg == p2.kg && p1.ring == p2.ring) {
          GemmP q = g;
          q.M = 2 * nb; q.nbatch = 1; q.a_batch_stride = 0; q.c_batch_stride = 0; q.epi.aux_batch_stride = 0;
          q.epi.sq_m_half = nb / 64;
          int cnt = 0;
          TRY(run_gemm(h, T_GEMM_DEC, T_RED_DEC, q, false, true, &cnt, 4.0 * 2 * nb * N));
          sqn = p2.sq_count;      // per path, as the two-batch form counts them
          merged = true;
        }
      }
      if (!merged)
      TRY(run_gemm(h, T_GEMM_DEC, T_RED_DEC, g, false, true, &sqn, 4.0 * 2 * nb * N));
    }
  } else {
    // rank out of rows: still open the optimizer step and contribute zeros to the collectives
    GANMF_LAUNCH(open_step_kernel, dim3(1), dim3(64), 0, h->st, h->scal, 0, aslot, h->cfg.d_lr);
    HIP_TRY(hipGetLastError());
  }
  if (dist) {
    MultiRed mr{};
    mr.count = 2; mr.out = h->scal;
    mr.e[0] = {h->sqp, sqn, S_SUM_REAL, 0};
    mr.e[1] = {h->sqp + sqn, sqn, S_SUM_FAKE, 0};
    {
      Scope s(h, T_MULTIRED, 0, 0);
      GANMF_LAUNCH(multi_reduce_kernel, dim3(S_SUM_FAKE + 1), dim3(256), 0, h->st, mr);      // block = destination slot
      HIP_TRY(hipGetLastError());
    }
    TRY(allreduce(h, h->scal + S_SUM_REAL, 2));
  }
  const DCoefP dc{h->scal, h->sqp, sqn, sqn, dist ? 1 : 0, h->cfg.m, nb, inv_bn, h->E, h->Es, h->lde, h->rs, parts};
  const long long dc_total = (long long)2 * nb * (h->lde / 4);
  // Single GPU, dE split along K: the dE GEMM only writes slabs (its row scale is applied when they are summed, inside the gWd
  // launch), so it does not depend on the hinge scalars and d_coef rides in ITS launch as extra blocks (de_dcoef_kernel).
  bool dcoef_done = false;
  GemmP gde{};
  GemmPlan pde;
  // (data-parallel runs take the same launch: the two sums d_coef needs were all-reduced just above)
  if (nb > 0 && (h->multi & 4) && (h->multi & 8)) {
    gde.A = h->Dl; gde.lda = h->ldN; gde.B = h->Wd.p; gde.ldb = h->ldN;
    gde.C = h->dE; gde.ldc = h->lde; gde.M = 2 * nb; gde.N = e; gde.K = N; gde.nbatch = 1;
    gde.epi.kind = EPI_ROWSCALE; gde.epi.rowscale = h->rs; gde.zero_page = h->zero_page;
    pde = gemm_plan(gde.M, gde.N, gde.K, 1, false, h->tune);
    pde.persist = 0;
    if (plan_is_f32_64_kg(pde, 4) && pde.nsplit > 1 && !(h->inkernel_reduce && pde.nsplit <= h->inlaunch_max)) {
      TRY(ensure_slab(h, gemm_slab_elems(pde, gde.M, gde.ldc, 1), 0));
      Scope s(h, T_DE_DCOEF, gemm_flops(gde.M, gde.N, gde.K), gemm_bytes(gde.M, gde.N, gde.K) + 8.0 * 2 * nb * e);
      GemmP q = gde;
      fill_plan(q, pde);
      q.C = h->slab; q.c_split_stride = (long long)gde.M * gde.ldc; q.c_batch_stride = (long long)gde.M * gde.ldc;
      const int ng = pde.tiles_m * pde.tiles_n * pde.nsplit;
      const int nd = h->dcoef_spread ? 0 : (int)std::max<long long>(1, std::min<long long>(32, (dc_total + 4095) / 4096));
      if (combined_x3(h)) GANMF_LAUNCH((de_dcoef_kernel<4, true>), dim3(ng + nd), dim3(1024), 0, h->st, q, dc, nd);
      else GANMF_LAUNCH(de_dcoef_kernel<4>, dim3(ng + nd), dim3(1024), 0, h->st, q, dc, nd);
      HIP_TRY(hipGetLastError());
      dcoef_done = true;
    }
  }
  if (!dcoef_done) {
    Scope s(h, T_DCOEF, 0, 8.0 * 2 * nb * e);
    const int grid = (int)std::max<long long>(1, std::min<long long>(128, (dc_total + 1023) / 1024));
    GANMF_LAUNCH(d_coef_kernel, dim3(grid), dim3(256), 0, h->st, dc);
    HIP_TRY(hipGetLastError());
  }
  if (nb > 0) {
    const bool regD = h->cfg.d_reg != 0.f;
    fused = h->fuse_adam && !dist;
    GemmTune ft;
    ft.tile = h->fused_tile; ft.ring = 2; ft.nsplit = 1;
    ft.mode = h->tune.mode != MFMA_AUTO ? h->tune.mode : h->fused_mode;
    ft.bk = h->fused_bk;
    // the data-parallel path runs the same two GEMMs with a plain store epilogue: same tile, split and arithmetic,
    // so that it stays bitwise equal to the fused single-GPU path (tests/test_gpu_parity.py, one-rank RCCL)
    const GemmTune* wg_tune = &ft;
    // sparse regime: the encoder-gradient GEMM runs over the generated rows only (A = F, B = dE_f, K = nb); the real rows' share
    // X^T . dE_r (and the bias row's column sums of dE_r) is added in its epilogue from the CSC matrix, before Adam / the store
    const int start = (int)(rows_dev - h->perm);      // position of this batch in the epoch permutation (pos[] is its inverse)
    auto sparse_gWe = [&](GemmP& g) {
      if (!h->sparse_d) return;
      g.A = h->XF + (size_t)nb * h->ldN; g.B = h->dE + (size_t)nb * h->lde; g.K = nb;
      g.epi.sp_rows = h->sp_rows; g.epi.sp_ld = h->lde; g.epi.sp_bias_row = N; g.epi.sp_bias_parts = CSC_BIAS_PARTS;
    };
    auto sparse_rows = [&]() -> int {      // S = X^T . dE_r (+ the bias parts): dE must be final
      if (!h->sparse_d) return 0;
      Scope s(h, T_DENSIFY, 0, 4.0 * ((double)h->nnz / std::max(h->U, 1) * nb * e + 2.0 * (N + CSC_BIAS_PARTS) * e));
      GANMF_LAUNCH(csc_rows_kernel, dim3(N + CSC_BIAS_PARTS), dim3(256), 0, h->st, h->csc_colptr, h->csc_rowidx, h->csc_val,
                   h->pos, start, nb, N, h->dE, h->lde, e, h->sp_rows);
      HIP_TRY(hipGetLastError());
      return 0;
    };
    RedP dE_red{};                  // single GPU: the slab sum of dE rides in the gWd launch (gWd does not read dE, gWe does)
    auto gemm_gWd = [&]() -> int {  // gWd_ext = (rs*[E|1])^T . Delta   -> rows 0..e-1 = gWd, row e = gbd
      GemmP g{};
      g.A = h->Es; g.lda = h->lde; g.B = h->Dl; g.ldb = h->ldN;
      g.C = h->Wd.g; g.ldc = h->ldN; g.M = e + 1; g.N = N; g.K = 2 * nb; g.epi.kind = EPI_STORE;
      g.a_scale = grad_scale(h, b_global);     // Es = rs (.) E carries the 2/(B.N) of the loss gradient
      if (fused) {
        g.epi.kind = EPI_ADAM; g.epi.adam_theta = h->Wd.p; g.epi.adam_m = h->Wd.m; g.epi.adam_v = h->Wd.v;
        g.epi.adam_alpha = h->scal + aslot; g.epi.adam_reg = h->cfg.d_reg;
        g.epi.sq_partials = regD ? regWd : nullptr;
      }
      return run_gemm(h, T_GEMM_GWD, T_RED_GWD, g, true, true, &regn[1], fused ? 24.0 * h->Wd.count() : 0, 0, wg_tune, nullptr,
                      nullptr, dE_red.part ? &dE_red : nullptr);
    };
    // Data-parallel: ONE order of collectives whatever this rank's row count or GEMM plans (ranks of a ragged step plan
    // differently; a rank without rows issues the same sequence on zero gradients): encoder first -- the next step needs We one
    // GEMM earlier than Wd.  dE (its slabs may exist already, de_dcoef_kernel) -> gWe_ext -> We's reduce-scatter / Adam slice /
    // all-gather on the side lane under the gWd_ext GEMM and the next step's row expansion + generator GEMM (joined before its
    // encode GEMM) -> gWd_ext -> Wd's behind them under that encode GEMM (joined before the decode GEMM; dE has read the old
    // decoder long before).
    if (dcoef_done) {   // the slabs are there already: hand their sum (+ row scale) to the gWd launch
      dE_red = RedP{};
      dE_red.part = h->slab; dE_red.nsplit = pde.nsplit; dE_red.out = gde.C; dE_red.ld = gde.ldc; dE_red.M = gde.M; dE_red.N = gde.N;
      dE_red.batch_stride = 0; dE_red.epi = gde.epi; dE_red.epi.sq_stride = GEMM_RED_GRID;      // (no partials: EPI_ROWSCALE)
      dE_red.split_stride = (long long)gde.M * gde.ldc;
    } else {  // dE = rs * (Delta . Wd^T)          (bias row e of Wd_ext is not part of this product; reads the OLD Wd)
      GemmP g{};
      g.A = h->Dl; g.lda = h->ldN; g.B = h->Wd.p; g.ldb = h->ldN;
      g.C = h->dE; g.ldc = h->lde; g.M = 2 * nb; g.N = e; g.K = N;
      g.epi.kind = EPI_ROWSCALE; g.epi.rowscale = h->rs;
      TRY(run_gemm(h, T_GEMM_DE, T_RED_DE, g, false, false, nullptr, 0, 0, nullptr, nullptr,
                   (!dist && (h->multi & 4)) ? &dE_red : nullptr));
    }
    // Weight-gradient GEMMs.  Single GPU: the gradient tile never leaves the CU -- the epilogue
    // applies TF-Adam to theta/m/v in place (after dE, which reads the old decoder).
    // Data-parallel: the gradients are stored, reduce-scattered, and each rank updates its slice (dp_update).
    bool wpair = false;
#ifdef GANMF_PERSIST_DIAG_BUILD
    if (!dist && fused && h->wgrad_stream && h->pl_XF && ft.mode == MFMA_BF16X3) {
      // Both weight-gradient products with TF-Adam in place as ONE persistent launch fed from bf16 x 3 planes (wgrad_stream.hpp).
      // In front of it, one launch: the slab sum of dE (+ its row scale; also files dE's planes) next to the planes of the three
      // operands that are final by now.
      const int K2 = 2 * nb;
      PlaneJobs js{};
      auto job = [&](const float* src, bf16raw* dst, long long ps, int rows, int ld) {
        js.j[js.count++] = PlaneJob{src, PlaneRef{dst, ps}, (long long)rows * ld / 4};
      };
      if (h->sparse_d) job(h->XF + (size_t)nb * h->ldN, h->pl_XF + (size_t)nb * h->ldN, h->ps_N, nb, h->ldN);      // (the real rows stay CSR)
      else job(h->XF, h->pl_XF, h->ps_N, K2, h->ldN);
      job(h->Dl, h->pl_Dl, h->ps_N, K2, h->ldN);
      job(h->Es, h->pl_Es, h->ps_e, K2, h->lde);
      RedPlanes rp{};
      RedP rr{};
      int nred = 0;
      long long n4 = 0;
      if (dE_red.part) {
        rr = dE_red; rp.pl = PlaneRef{h->pl_dE, h->ps_e}; rp.on = 1;
        nred = h->red_elems > 0 ? red_grid(h, dE_red.M, dE_red.N, dE_red.nsplit) : GEMM_RED_GRID;
        rr.epi.sq_stride = nred;
      } else job(h->dE, h->pl_dE, h->ps_e, K2, h->lde);
      for (int i = 0; i < js.count; ++i) n4 += js.j[i].n4;
      const int nsp = (int)std::max<long long>(1, std::min<long long>(1024, (n4 + 1023) / 1024));
      {
        Scope s(h, T_RED_DE, 0, (dE_red.part ? 4.0 * (dE_red.nsplit + 1) * dE_red.M * dE_red.N : 0.0) + 10.0 * 4.0 * (double)n4);
        GANMF_LAUNCH(presplit_red_kernel, dim3(nred + nsp), dim3(256), 0, h->st, rr, rp, nred, js);
        HIP_TRY(hipGetLastError());
      }
      TRY(sparse_rows());
      WgsP w{};
      w.zero_page = h->zero_page;
      w.dump = h->wgs_dump;
      WgsProd& w0 = w.g[0];
      w0.a_pl = h->pl_Es; w0.a_ps = h->ps_e; w0.lda = h->lde; w0.b_pl = h->pl_Dl; w0.b_ps = h->ps_N; w0.ldb = h->ldN; w0.ldc = h->ldN;
      w0.M = e + 1; w0.N = N; w0.K = K2;
      w0.epi.kind = EPI_ADAM; w0.epi.adam_theta = h->Wd.p; w0.epi.adam_m = h->Wd.m; w0.epi.adam_v = h->Wd.v;
      w0.epi.adam_alpha = h->scal + aslot; w0.epi.adam_reg = h->cfg.d_reg; w0.epi.sq_partials = regD ? regWd : nullptr;
      WgsProd& w1 = w.g[1];
      const size_t r1 = h->sparse_d ? (size_t)nb : 0;      // sparse regime: the generated rows only, the real rows' share comes from the CSC matrix
      w1.a_pl = h->pl_XF + r1 * h->ldN; w1.a_ps = h->ps_N; w1.lda = h->ldN; w1.b_pl = h->pl_dE + r1 * h->lde; w1.b_ps = h->ps_e; w1.ldb = h->lde;
      w1.ldc = h->lde; w1.M = N + 1; w1.N = e; w1.K = h->sparse_d ? nb : K2;
      w1.epi.kind = EPI_ADAM; w1.epi.adam_theta = h->We.p; w1.epi.adam_m = h->We.m; w1.epi.adam_v = h->We.v;
      w1.epi.adam_alpha = h->scal + aslot; w1.epi.adam_reg = h->cfg.d_reg; w1.epi.sq_partials = regD ? regWe : nullptr;
      if (h->sparse_d) { w1.epi.sp_rows = h->sp_rows; w1.epi.sp_ld = h->lde; w1.epi.sp_bias_row = N; w1.epi.sp_bias_parts = CSC_BIAS_PARTS; }
      for (int i = 0; i < 2; ++i) { w.g[i].tiles_m = (w.g[i].M + WGS_BM - 1) / WGS_BM; w.g[i].tiles_n = (w.g[i].N + WGS_BN - 1) / WGS_BN; }
      w.tiles0 = w0.tiles_m * w0.tiles_n;
      w.tiles_total = w.tiles0 + w1.tiles_m * w1.tiles_n;
      regn[1] = w.tiles0; regn[0] = w.tiles_total - w.tiles0;
      const int grid = std::min(GEMM_CUS, round_up(w.tiles_total, 8));
      {
        const int key[5] = {w0.tiles_m, w0.tiles_n, w1.tiles_m, w1.tiles_n, grid};
        if (memcmp(key, h->wgs_key, sizeof key) != 0) {
          const int tms[2] = {w0.tiles_m, w1.tiles_m}, tns[2] = {w0.tiles_n, w1.tiles_n};
          std::vector<int> tab;
          const int rounds = wgs_build_schedule(tms, tns, 2, grid, tab);
          HIP_TRY(hipStreamSynchronize(h->st));
          if (tab.size() > h->wgs_table_cap) {
            hipFree(h->wgs_table); h->wgs_table = nullptr;
            HIP_TRY(hipMalloc((void**)&h->wgs_table, tab.size() * sizeof(int)));
            h->wgs_table_cap = tab.size();
          }
          HIP_TRY(hipMemcpy(h->wgs_table, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
          memcpy(h->wgs_key, key, sizeof key);
          h->wgs_rounds = rounds;
        }
        w.table = h->wgs_table; w.rounds = h->wgs_rounds;
      }
      w.diag = env_int("GANMF_WGS_DIAG", 0);
      static unsigned long long* wst = nullptr;
      static int wst_n = 0;
      const bool stamping = env_int("GANMF_WGS_STAMPS", 0) != 0 && wst_n < 40;
      if (stamping) {
        if (!wst) HIP_TRY(hipMalloc((void**)&wst, (size_t)GEMM_CUS * 64 * 8));
        HIP_TRY(hipMemsetAsync(wst, 0, (size_t)GEMM_CUS * 64 * 8, h->st));
        w.stamps = wst;
      }
      {
        Scope s(h, T_WPAIR, gemm_flops(w0.M, w0.N, w0.K) + gemm_flops(w1.M, w1.N, w1.K),
                // operands once (as planes: 6 bytes per element) + the six Adam streams; the gradients themselves never reach HBM
                6.0 * ((double)w0.K * (w0.M + w0.N) + (double)w1.K * (w1.M + w1.N)) + 24.0 * ((double)w0.M * w0.N + (double)w1.M * w1.N));
        GANMF_LAUNCH(wgrad_stream_kernel, dim3(grid), dim3(1024), 0, h->st, w);
        HIP_TRY(hipGetLastError());
      }
      if (stamping && ++wst_n == 40) {      // (a warm launch)
        HIP_TRY(hipStreamSynchronize(h->st));
        std::vector<unsigned long long> hs((size_t)grid * 64);
        HIP_TRY(hipMemcpy(hs.data(), wst, hs.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < grid; ++b) if (hs[(size_t)b * 64]) t0 = std::min(t0, hs[(size_t)b * 64]);
        auto med = [&](int role, int i) {
          std::vector<double> v;
          for (int b = 0; b < grid; ++b) { const unsigned long long x = hs[((size_t)b * 2 + role) * 32 + i]; if (x) v.push_back((double)(x - t0) * 0.01); }
          if (v.empty()) return -1.0;
          std::sort(v.begin(), v.end());
          return v[v.size() / 2];
        };
        fprintf(stderr, "[wgs stamps] diag %d, %d workgroups, medians in us after the first GEMM-wave entry\n", w.diag, grid);
        for (int role = 0; role < 2; ++role) {
          fprintf(stderr, "  %s: start %.2f |", role ? "Adam" : "GEMM", med(role, 0));
          for (int r = 0; r < 6; ++r) fprintf(stderr, " r%d: work done %.2f, past Y %.2f, past X %.2f |", r, med(role, 1 + 3 * r), med(role, 2 + 3 * r), med(role, 3 + 3 * r));
          fprintf(stderr, "\n   detail:");
          for (int i = 19; i < 32; ++i) fprintf(stderr, " %.2f", med(role, i));
          fprintf(stderr, "\n");
        }
      }
      wpair = true;
    }
#endif
    if (!wpair && !dist && fused && (h->multi & 16) && dE_red.part) {
      // both weight-gradient products in ONE launch (wgrad_pair_kernel); the slab sum of dE, which gWe reads, gets its own
      // launch in front
      GemmP g0{}, g1{};
      g0.A = h->Es; g0.lda = h->lde; g0.B = h->Dl; g0.ldb = h->ldN;
      g0.C = h->Wd.g; g0.ldc = h->ldN; g0.M = e + 1; g0.N = N; g0.K = 2 * nb; g0.nbatch = 1; g0.zero_page = h->zero_page;
      g0.a_scale = grad_scale(h, b_global);
      g0.epi.kind = EPI_ADAM; g0.epi.adam_theta = h->Wd.p; g0.epi.adam_m = h->Wd.m; g0.epi.adam_v = h->Wd.v;
      g0.epi.adam_alpha = h->scal + aslot; g0.epi.adam_reg = h->cfg.d_reg; g0.epi.sq_partials = regD ? regWd : nullptr;
      g1.A = h->XF; g1.lda = h->ldN; g1.B = h->dE; g1.ldb = h->lde;
      g1.C = h->We.g; g1.ldc = h->lde; g1.M = N + 1; g1.N = e; g1.K = 2 * nb; g1.nbatch = 1; g1.zero_page = h->zero_page;
      g1.b_scale = grad_scale(h, b_global);
      g1.epi.kind = EPI_ADAM; g1.epi.adam_theta = h->We.p; g1.epi.adam_m = h->We.m; g1.epi.adam_v = h->We.v;
      g1.epi.adam_alpha = h->scal + aslot; g1.epi.adam_reg = h->cfg.d_reg; g1.epi.sq_partials = regD ? regWe : nullptr;
      sparse_gWe(g1);
      GemmPlan p0 = gemm_plan(g0.M, g0.N, g0.K, 1, regD, ft, true), p1 = gemm_plan(g1.M, g1.N, g1.K, 1, regD, ft, true);
      auto staged = [](const GemmPlan& pl) { return pl.mode == MFMA_BF16X3 && pl.tile == 64 && pl.bk == 32 && pl.nsplit == 1; };
      if (staged(p0) && staged(p1)) {
        {
          Scope s(h, T_RED_DE, 0, 4.0 * (dE_red.nsplit + 1) * dE_red.M * dE_red.N);
          GANMF_LAUNCH(splitk_reduce_kernel, dim3(h->red_elems > 0 ? red_grid(h, dE_red.M, dE_red.N, dE_red.nsplit) : GEMM_RED_GRID, 1), dim3(256), 0, h->st, dE_red);
          HIP_TRY(hipGetLastError());
        }
        TRY(sparse_rows());
        fill_plan(g0, p0);
        fill_plan(g1, p1);
        g0.n_fastest = g1.n_fastest = (h->adam_nfast & 1);
#ifdef GANMF_PERSIST_DIAG_BUILD
        // diagnostic build only (make DIAG=1; wrong results): GANMF_WGRAD_DIAG=1 empties the K range, i.e. the launch becomes its
        // tile-wise Adam pass on a zero gradient -- how long do the twelve Adam streams take in THIS access pattern and occupancy
        if (env_int("GANMF_WGRAD_DIAG", 0) & 1) { g0.K = 0; g1.K = 0; g0.k_per_split = g1.k_per_split = 0; }
        g0.diag = g1.diag = env_int("GANMF_WGRAD_DIAG", 0);      // bit 1 (2): no 3-way split, bit 2 (4): 1/NC of the MFMAs, bit 3 (8): no Adam streams
#endif
        regn[1] = p0.sq_count; regn[0] = p1.sq_count;
        const int n0 = p0.tiles_m * p0.tiles_n, n1 = p1.tiles_m * p1.tiles_n;
        Scope s(h, T_WPAIR, gemm_flops(g0.M, g0.N, g0.K) + gemm_flops(g1.M, g1.N, g1.K),
                // operands once + the six Adam streams; the gradients themselves never reach HBM
                4.0 * ((double)g0.K * (g0.M + g0.N) + (double)g1.K * (g1.M + g1.N)) + 24.0 * ((double)g0.M * g0.N + (double)g1.M * g1.N));
#ifdef GANMF_PERSIST_DIAG_BUILD
        // diagnostic build, GANMF_WGRAD_STAMPS=1: the launch's workgroup timelines (64 stamps each, gemm_bf16s.hpp), summarised for the
        // first launches: lifetimes, and per K-tile the five phases averaged over the workgroups
        static unsigned long long* wst = nullptr;
        static int wst_printed = 0;
        const bool stamping = env_int("GANMF_WGRAD_STAMPS", 0) != 0 && wst_printed < 3;
        if (stamping) {
          if (!wst) HIP_TRY(hipMalloc((void**)&wst, (size_t)(n0 + n1) * 64 * 8));
          HIP_TRY(hipMemsetAsync(wst, 0, (size_t)(n0 + n1) * 64 * 8, h->st));
          g0.stamps = wst; g1.stamps = wst + (size_t)n0 * 64;
        }
#endif
        GANMF_LAUNCH(wgrad_pair_kernel, dim3(n0 + n1), dim3(256), 0, h->st, g0, g1);
        HIP_TRY(hipGetLastError());
#ifdef GANMF_PERSIST_DIAG_BUILD
        if (stamping && ++wst_printed >= 2) {      // (the second launch: warm)
          HIP_TRY(hipStreamSynchronize(h->st));
          const int nw = n0 + n1;
          std::vector<unsigned long long> hs((size_t)nw * 64);
          HIP_TRY(hipMemcpy(hs.data(), wst, hs.size() * 8, hipMemcpyDeviceToHost));
          unsigned long long t0 = ~0ull, t1 = 0;
          for (int b = 0; b < nw; ++b) { t0 = std::min(t0, hs[(size_t)b * 64]); t1 = std::max(t1, hs[(size_t)b * 64 + 63]); }
          auto med = [&](auto f) { std::vector<double> v(nw); for (int b = 0; b < nw; ++b) v[b] = f(b) * 0.01; std::sort(v.begin(), v.end()); return std::make_pair(v[nw / 2], v[(size_t)(nw * 0.9)]); };
          auto pr = [&](const char* name, std::pair<double, double> m) { fprintf(stderr, "  %-44s median %6.2f  p90 %6.2f us\n", name, m.first, m.second); };
          fprintf(stderr, "[wgrad stamps] %d + %d workgroups, first entry -> last exit %.2f us\n", n0, n1, (t1 - t0) * 0.01);
          pr("entry after the first entry", med([&](int b) { return (double)(hs[(size_t)b * 64] - t0); }));
          pr("entry -> first K-tile's loads issued", med([&](int b) { return (double)(hs[(size_t)b * 64 + 1] - hs[(size_t)b * 64]); }));
          pr("K loop (8 K-tiles)", med([&](int b) { return (double)(hs[(size_t)b * 64 + 62] - hs[(size_t)b * 64 + 1]); }));
          pr("epilogue (Adam streams) + store drain", med([&](int b) { return (double)(hs[(size_t)b * 64 + 63] - hs[(size_t)b * 64 + 62]); }));
          for (int it = 0; it < 8; ++it) {
            const int o = 1 + 5 * it;
            fprintf(stderr, "  K-tile %d:", it);
            auto m1 = med([&](int b) { return (double)(hs[(size_t)b * 64 + o + 1] - hs[(size_t)b * 64 + o]); });
            auto m2 = med([&](int b) { return (double)(hs[(size_t)b * 64 + o + 2] - hs[(size_t)b * 64 + o + 1]); });
            fprintf(stderr, " fragments + MFMA issue %5.2f (p90 %5.2f)  barrier %5.2f (%5.2f)", m1.first, m1.second, m2.first, m2.second);
            if (it < 7) {
              auto m3 = med([&](int b) { return (double)(hs[(size_t)b * 64 + o + 3] - hs[(size_t)b * 64 + o + 2]); });
              auto m4 = med([&](int b) { return (double)(hs[(size_t)b * 64 + o + 4] - hs[(size_t)b * 64 + o + 3]); });
              auto m5 = med([&](int b) { return (double)(hs[(size_t)b * 64 + o + 5] - hs[(size_t)b * 64 + o + 4]); });
              fprintf(stderr, "  wait for loads + split + plane stores %5.2f (%5.2f)  barrier %5.2f (%5.2f)  next loads issued %5.2f", m3.first, m3.second, m4.first, m4.second, m5.first);
            }
            fprintf(stderr, " us\n");
          }
        }
#endif
        wpair = true;
      }
    }
    if (dist && dE_red.part) {     // (dcoef_done: the slabs of dE are waiting for their sum)
      Scope s(h, T_RED_DE, 0, 4.0 * (dE_red.nsplit + 1) * dE_red.M * dE_red.N);
      GANMF_LAUNCH(splitk_reduce_kernel, dim3(h->red_elems > 0 ? red_grid(h, dE_red.M, dE_red.N, dE_red.nsplit) : GEMM_RED_GRID, 1), dim3(256), 0, h->st, dE_red);
      HIP_TRY(hipGetLastError());
      dE_red.part = nullptr;      // (summed: nothing to attach to the gWd launch)
    }
    if (!dist && !wpair) TRY(gemm_gWd());
    if (!wpair) {  // gWe_ext = [X;F | 1]^T . dE       -> rows 0..N-1 = gWe, row N = gbe
      TRY(sparse_rows());
      GemmP g{};
      g.A = h->XF; g.lda = h->ldN; g.B = h->dE; g.ldb = h->lde;
      g.C = h->We.g; g.ldc = h->lde; g.M = N + 1; g.N = e; g.K = 2 * nb; g.epi.kind = EPI_STORE;
      g.b_scale = grad_scale(h, b_global);     // dE
      if (fused) {
        g.epi.kind = EPI_ADAM; g.epi.adam_theta = h->We.p; g.epi.adam_m = h->We.m; g.epi.adam_v = h->We.v;
        g.epi.adam_alpha = h->scal + aslot; g.epi.adam_reg = h->cfg.d_reg;
        g.epi.sq_partials = regD ? regWe : nullptr;
      }
      sparse_gWe(g);
      const bool armed = dist && fork_arm(h);
      TRY(run_gemm(h, T_GEMM_GWE, T_RED_GWE, g, true, true, &regn[0], fused ? 24.0 * h->We.count() : 0, 0, wg_tune));
      if (dist) TRY(fork_wait(h, armed));       // the side lane waits for gWe_ext
    }
    if (dist) {
      TRY(dp_update(h, T_ADAM_D, h->We, aslot, h->cfg.d_reg, regD ? regWe : nullptr, 1));
      TRY(dp_mark(h, PEND_WE));
      const bool armed = fork_arm(h);
      TRY(gemm_gWd());
      TRY(fork_wait(h, armed));       // ... and for gWd_ext
      TRY(dp_update(h, T_ADAM_D, h->Wd, aslot, h->cfg.d_reg, regD ? regWd : nullptr, 1));
      TRY(dp_mark(h, PEND_WD));
      (void)regn; (void)parts;
      return 0;
    }
  } else {
    // (ranks with rows join inside step_front; the previous step's encoder update may still be reducing We.g on the side lane)
    TRY(dp_join(h));
    HIP_TRY(hipMemsetAsync(h->gD, 0, h->gD_elems * sizeof(float), h->st));
    if (dist) {   // same collective order as the ranks that have rows: encoder, then decoder
      const bool regD = h->cfg.d_reg != 0.f;
      TRY(lane_fork(h));
      TRY(dp_update(h, T_ADAM_D, h->We, aslot, h->cfg.d_reg, regD ? regWe : nullptr, 1));
      TRY(dp_mark(h, PEND_WE));
      TRY(dp_update(h, T_ADAM_D, h->Wd, aslot, h->cfg.d_reg, regD ? regWd : nullptr, 1));
      TRY(dp_mark(h, PEND_WD));
    }
  }
  const bool reg = h->cfg.d_reg != 0.f;
  if (!dist && !fused) {
    TRY(adam_dense(h, T_ADAM_D, h->We, h->We.g, aslot, h->cfg.d_reg, reg ? regWe : nullptr));
    TRY(adam_dense(h, T_ADAM_D, h->Wd, h->Wd.g, aslot, h->cfg.d_reg, reg ? regWd : nullptr));
  }
  (void)regn; (void)parts;
  return 0;   // the sum(theta^2) partials are reduced once per epoch (finish_parts_kernel)
}

// Generator parameter update shared by GANMF and DisGANMF: gUb = dF.V (reads the OLD V), gV = dF^T.Ub,
// Adam on V (fused into the gV GEMM epilogue on a single GPU) and the all-rows Adam on U.
// *regn_v = number of sum(V^2) partials written (when g_reg != 0).
int gen_update(ganmf_handle* h, int nb, int start, int b_global, int* regn_v, float* reg_u, float* reg_v) {
  const int N = h->N, k = h->k;
  const bool reg = h->cfg.g_reg != 0.f;
  const bool dist = h->has_comm;
  const bool fused = h->fuse_adam && !dist && nb > 0;
  *regn_v = ADAM_GRID;
  SlabRef gub{h->gUb, 1, 0};
  auto gemm_gUb = [&]() -> int {  // gUb = dF . V     (reads the OLD V)
    GemmP g{};
    g.A = h->dF; g.lda = h->ldN; g.B = h->V.p; g.ldb = h->ldk;
    g.C = h->gUb; g.ldc = h->ldk; g.M = nb; g.N = k; g.K = N; g.epi.kind = EPI_STORE;
    g.a_scale = grad_scale(h, b_global);      // dF
    return run_gemm(h, T_GEMM_GUB, T_RED_GUB, g, false, true, nullptr, 0, 0, nullptr, h->defer_gub ? &gub : nullptr);   // slabs summed by adam_rows_kernel
  };
  auto gemm_gV = [&]() -> int {   // gV = dF^T . Ub
    GemmP g{};
    g.A = h->dF; g.lda = h->ldN; g.B = h->Ub; g.ldb = h->ldk;
    g.C = h->V.g; g.ldc = h->ldk; g.M = N; g.N = k; g.K = nb; g.epi.kind = EPI_STORE;
    g.a_scale = grad_scale(h, b_global);      // dF
    if (fused) {
      g.epi.kind = EPI_ADAM; g.epi.adam_theta = h->V.p; g.epi.adam_m = h->V.m; g.epi.adam_v = h->V.v;
      g.epi.adam_alpha = h->scal + S_ALPHA_G; g.epi.adam_reg = h->cfg.g_reg;
      g.epi.sq_partials = reg ? reg_v : nullptr;
    }
    GemmTune ft;
    ft.tile = 64; ft.ring = 2; ft.nsplit = 1;
    ft.mode = h->tune.mode != MFMA_AUTO ? h->tune.mode : h->fused_mode;
    ft.bk = h->fused_bk;
    // the half-depth staged split-bf16 kernel of the other fused-Adam GEMMs once the output has at least two 64 x 64 tiles per
    // CU (C4 width, N = 50 000: 136 -> 102 us, +2 .. 5 % steps/s); below that the fp32 ring kernel (C2: 17.2 vs 18.1 us, and it
    // pairs with gUb in one launch).  GANMF_GV_STAGED = 0 / 1 forces either.
    const long long gv_tiles = (long long)((N + 63) / 64) * ((k + 63) / 64);
    const bool staged_gv = env_int("GANMF_GV_STAGED", gv_tiles >= 2 * GEMM_CUS ? 1 : 0) != 0;
    TRY(run_gemm(h, T_GEMM_GV, T_RED_GV, g, true, true, regn_v, fused ? 24.0 * h->V.count() : 0, 0, staged_gv ? &ft : nullptr));
    if (!fused) *regn_v = ADAM_GRID;
    return 0;
  };
  // gUb and gV in ONE launch (pair_kernel): both read dF.  Single GPU: gUb reads the old V while gV's Adam epilogue writes the
  // new V into the second buffer, swapped in afterwards.  Data-parallel: gV is stored (it must be reduced before its Adam), V is
  // not touched inside the launch.  Only when both are planned onto the 16-wave fp32 ring kernel.
  bool paired = false, pair_armed = false;
  if (nb > 0 && (fused || dist) && (h->multi & 2) && h->defer_gub && h->V_alt) {
    if (dist) TRY(dp_join(h));
    GemmP g0{}, g1{};
    g0.A = h->dF; g0.lda = h->ldN; g0.B = h->V.p; g0.ldb = h->ldk;
    g0.C = h->gUb; g0.ldc = h->ldk; g0.M = nb; g0.N = k; g0.K = N; g0.epi.kind = EPI_STORE; g0.nbatch = 1;
    g0.zero_page = h->zero_page; g0.a_scale = grad_scale(h, b_global);
    g1.A = h->dF; g1.lda = h->ldN; g1.B = h->Ub; g1.ldb = h->ldk;
    g1.C = h->V.g; g1.ldc = h->ldk; g1.M = N; g1.N = k; g1.K = nb; g1.nbatch = 1;
    g1.zero_page = h->zero_page; g1.a_scale = grad_scale(h, b_global);
    g1.epi.kind = EPI_STORE;
    if (fused) {
      g1.epi.kind = EPI_ADAM; g1.epi.adam_theta = h->V.p; g1.epi.adam_theta_out = h->V_alt; g1.epi.adam_m = h->V.m; g1.epi.adam_v = h->V.v;
      g1.epi.adam_alpha = h->scal + S_ALPHA_G; g1.epi.adam_reg = h->cfg.g_reg;
      g1.epi.sq_partials = reg ? reg_v : nullptr;
    }
    GemmPlan p0 = gemm_plan(g0.M, g0.N, g0.K, 1, false, h->tune);
    GemmPlan p1 = gemm_plan(g1.M, g1.N, g1.K, 1, g1.epi.sq_partials != nullptr, h->tune, true);
    if (plan_is_f32_64_kg(p0, 4) && plan_is_f32_64_kg(p1, 4) && p1.nsplit == 1 &&
        reduce_groups((long long)g0.M * ((g0.N + 3) >> 2), p0.nsplit) == 1) {
      if (p0.nsplit > 1) {
        TRY(ensure_slab(h, gemm_slab_elems(p0, g0.M, g0.ldc, 1), 1));
        g0.C = h->slab2; g0.c_split_stride = (long long)g0.M * g0.ldc;
        gub = SlabRef{h->slab2, p0.nsplit, (long long)g0.M * g0.ldc};
      }
      g0.c_batch_stride = (long long)g0.M * g0.ldc;
      fill_plan(g0, p0);
      fill_plan(g1, p1);
      if (fused && (h->adam_nfast & 2)) g1.n_fastest = 1;
      if (fused) *regn_v = p1.sq_count;
      const int n0 = p0.tiles_m * p0.tiles_n * p0.nsplit, n1 = p1.tiles_m * p1.tiles_n;
      {
        Scope s(h, T_PAIR, gemm_flops(g0.M, g0.N, g0.K) + gemm_flops(g1.M, g1.N, g1.K),
                gemm_bytes(g0.M, g0.N, g0.K) + gemm_bytes(g1.M, g1.N, g1.K) + (fused ? 24.0 * h->V.count() : 0));
        pair_armed = dist && fork_arm(h);
        if (h->pair_ring == 2) GANMF_LAUNCH((pair_kernel<4, 2>), dim3(n0 + n1), dim3(1024), 0, h->st, g0, g1);
        else GANMF_LAUNCH((pair_kernel<4, 3>), dim3(n0 + n1), dim3(1024), 0, h->st, g0, g1);
        HIP_TRY(hipGetLastError());
      }
      if (fused) std::swap(h->V.p, h->V_alt);
      paired = true;
    }
  }
  if (dist && paired) {
    // the whole update of V on the side lane, under the all-rows Adam pass over U (which reads neither V nor its gradient)
    TRY(fork_wait(h, pair_armed));
    TRY(dp_update(h, T_ADAM_V, h->V, S_ALPHA_G, h->cfg.g_reg, reg ? reg_v : nullptr, 1));
  } else if (dist) {
    // data-parallel, separate launches: gV first; its reduce-scatter runs on the side lane under gUb (which reads the OLD V),
    // the Adam slice and the all-gather of V under the all-rows Adam pass over U
    TRY(dp_join(h));
    if (nb > 0) TRY(gemm_gV());
    else HIP_TRY(hipMemsetAsync(h->V.g, 0, h->V.cap * sizeof(float), h->st));
    TRY(lane_fork(h));
    TRY(reduce_scatter(h, h->V.g, h->V.cap, 1));
    if (nb > 0) TRY(gemm_gUb());
    TRY(lane_fork(h));
    const size_t slice = h->V.cap / (size_t)h->cfg.world_size, off = (size_t)h->cfg.rank * slice;
    TRY(adam_dense(h, T_ADAM_V, h->V, h->V.g, S_ALPHA_G, h->cfg.g_reg, reg ? reg_v : nullptr, 1, off, slice));
    TRY(all_gather(h, h->V.p, h->V.cap, 1));
  } else if (nb > 0) {
    if (!paired) {
      TRY(gemm_gUb());
      TRY(gemm_gV());      // fused: Adam(V) in the epilogue, after gUb has read the old V
    }
  } else {
    HIP_TRY(hipMemsetAsync(h->V.g, 0, h->V.padded() * sizeof(float), h->st));
  }
  {
    Scope s(h, T_ADAM_U, 0, 24.0 * h->Ue.count());
    GANMF_LAUNCH(adam_rows_kernel, dim3(ADAM_GRID), dim3(256), 0, h->st, h->Ue.p, h->Ue.m, h->Ue.v, gub.p,
                       gub.nsplit, gub.split_stride, h->pos, start, nb, h->U, h->ldk, h->scal, S_ALPHA_G, h->cfg.g_reg,
                       reg ? reg_u : nullptr);
    HIP_TRY(hipGetLastError());
  }
  if (dist) TRY(lane_join(h));      // V is read by the very next kernel of the next step (generator GEMM)
  else if (!fused) TRY(adam_dense(h, T_ADAM_V, h->V, h->V.g, S_ALPHA_G, h->cfg.g_reg, reg ? reg_v : nullptr));
  return 0;
}

// One generator update (GANMF.py:133-135,139,200-201).  `start` = position of the batch in the
// epoch permutation (adam_rows_kernel finds batch rows through pos[]).
int g_step(ganmf_handle* h, const int* rows_dev, int nb, int start, int b_global, float* parts, float* arena) {
  // this step's arena slot: seg0 = sum Delta_f^2, seg1 = sum (Ef-Er)^2, seg2 = sum U^2, seg3 = sum V^2 partials
  const size_t cap = h->reg_cap;
  const int N = h->N, e = h->e;
  const float alpha = h->cfg.recon_coefficient;
  const float inv_bn = 1.0f / ((float)b_global * (float)N);
  int sqn = 0, fmn = 0;
  if (nb > 0) {
    TRY(step_front(h, rows_dev, nb, 1));
    TRY(dp_join(h, PEND_WD));
    {  // Delta_f = [Ef|1].Wd_ext - F, sum of squares
      GemmP g{};
      g.A = h->E + (size_t)nb * h->lde; g.lda = h->lde; g.B = h->Wd.p; g.ldb = h->ldN;
      g.C = h->Dl; g.ldc = h->ldN; g.M = nb; g.N = N; g.K = e + 1;
      g.epi.kind = EPI_SUB_AUX_SQ; g.epi.aux = h->XF + (size_t)nb * h->ldN; g.epi.ldaux = h->ldN;
      g.epi.sq_partials = arena;
      TRY(run_gemm(h, T_GEMM_DEC, T_RED_DEC, g, false, true, &sqn, 4.0 * nb * N));
    }
    // host constants: rsG = (1-alpha)*2/(B*N) ; cfm = alpha*2/(B*e)
    const float rsv = (1.0f - alpha) * (2.0f * inv_bn);
    const float cfm = alpha * 2.0f / ((float)b_global * (float)e);
    {  // dE = rsG*(Delta_f . Wd^T) + cfm*(Ef - Er) ; FM partials
      GemmP g{};
      g.A = h->Dl; g.lda = h->ldN; g.B = h->Wd.p; g.ldb = h->ldN;
      g.C = h->dE; g.ldc = h->lde; g.M = nb; g.N = e; g.K = N;
      g.epi.kind = EPI_G_DE; g.epi.c = rsv; g.epi.cfm = cfm;
      g.epi.er = h->E; g.epi.ef = h->E + (size_t)nb * h->lde; g.epi.sq_partials = arena + cap;
      TRY(run_gemm(h, T_GEMM_DE, T_RED_DE, g, false, false, &fmn));
    }
    {  // dF = dE . We^T - rsG*Delta_f      (MSE gradient reaches F through both arguments)
      GemmP g{};
      g.A = h->dE; g.lda = h->lde; g.B = h->We.p; g.ldb = h->lde;
      g.C = h->dF; g.ldc = h->ldN; g.M = nb; g.N = N; g.K = e;
      g.epi.kind = EPI_SUB_SCALED_AUX; g.epi.c = rsv; g.epi.aux = h->Dl; g.epi.ldaux = h->ldN;
      g.a_scale = grad_scale(h, b_global);     // dE
      TRY(run_gemm(h, T_GEMM_DF, T_RED_DF, g, false, false, nullptr, 4.0 * nb * N));
    }
  } else {
    GANMF_LAUNCH(open_step_kernel, dim3(1), dim3(64), 0, h->st, h->scal, 1, S_ALPHA_G, h->cfg.g_lr);
    HIP_TRY(hipGetLastError());
  }
  const bool reg = h->cfg.g_reg != 0.f;
  int regn_v = ADAM_GRID;
  TRY(gen_update(h, nb, start, b_global, &regn_v, arena + 2 * cap, arena + 3 * cap));
  (void)parts; (void)reg; (void)sqn; (void)fmn;
  return 0;
}

// =================================================================================================
// DisGANMF (GANRec/DisGANMF.py:57-79,110-140): binary MLP discriminator on [float(uid) | profile].
// Layer-0 input = XF with the ones column at N and float(uid) at column N+1; W_0_ext rows follow the
// same order (profile rows, bias row, uid row), so the uid term is an exact fp32 rank-1 part of the
// same GEMM.  Hidden outputs carry a ones column at e (bias folding as in GANMF).
// =================================================================================================
int dis_forward(ganmf_handle* h, const int* rows_dev, int nb, int which) {
  const int N = h->N, k = h->k, e = h->e;
  // X rows with the float(uid) column, Ub, F = Ub . V^T   (DisGANMF.py:59,77-78)
  TRY(rows_and_generator(h, rows_dev, nb, which, which ? S_ALPHA_G : S_ALPHA_D, N + 1, (int)h->cfg.row_offset));
  (void)k;
  for (int l = 0; l < h->L; ++l) {  // a_l = act([a_{l-1} | 1 (| uid)] . W_l_ext)   (DisGANMF.py:60-62)
    GemmP g{};
    g.A = l == 0 ? h->XF : h->Al[l - 1]; g.lda = l == 0 ? h->ldN : h->lde;
    g.B = h->Wl[l].p; g.ldb = h->lde;
    g.C = h->Al[l]; g.ldc = h->lde; g.M = 2 * nb; g.N = e; g.K = l == 0 ? N + 2 : e + 1;
    g.epi.kind = EPI_ACT; g.epi.act = h->act;
    if (l == 0 && low_precision(h)) {
      // float(uid) (up to 6039: not representable in 8 or 11 bits) stays fp32: column N+1 of the input leaves the K range
      // and comes back as a rank-1 term of the epilogue, uid[m] * W_0_ext[N+1, n]
      g.K = N + 1;
      g.epi.r1_u = h->XF + (N + 1); g.epi.r1_ld = h->ldN;
      g.epi.r1_w = h->Wl[0].p + (size_t)(N + 1) * h->lde;
    }
    TRY(run_gemm(h, T_DIS_FWD, T_RED_DIS_FWD, g, false, true));
  }
  return 0;
}

// dz_{l-1} = (dz_l . W_l[0:e]^T) * act'(a_{l-1}) for rows [row0, row0+nrows); returns the dz_0 buffer
int dis_backprop_hidden(ganmf_handle* h, int row0, int nrows, bool param_grads, int b_global, float** dz0_out) {
  const float gsc = grad_scale(h, b_global);     // dz carries the 1/B of the mean cross-entropy times an output weight
  const int N = h->N, e = h->e;
  float* cur = h->dz0;   // dz_{L-1} was written here by dis_dz_top_kernel
  float* nxt = h->dz1;
  auto backward = [&](int l) -> int {      // dz_{l-1} = (dz_l . W_l[0:e]^T) * act'(a_{l-1}): cur -> nxt
    GemmP g{};
    g.A = cur + (size_t)row0 * h->lde; g.lda = h->lde; g.B = h->Wl[l].p; g.ldb = h->lde;
    g.C = nxt + (size_t)row0 * h->lde; g.ldc = h->lde; g.M = nrows; g.N = e; g.K = e;
    g.epi.kind = EPI_MUL_ACTGRAD; g.epi.act = h->act;
    g.epi.aux = h->Al[l - 1] + (size_t)row0 * h->lde; g.epi.ldaux = h->lde;
    g.a_scale = gsc;
    return run_gemm(h, T_DIS_BWD, T_RED_DIS_BWD, g, false, false);
  };
  for (int l = h->L - 1; l >= 0; --l) {
    if (param_grads) {  // gW_l_ext = [a_{l-1} | 1 (| uid)]^T . dz_l  (all 2B rows: row0 = 0)
      GemmP g{};
      g.A = l == 0 ? h->XF : h->Al[l - 1]; g.lda = l == 0 ? h->ldN : h->lde;
      g.B = cur; g.ldb = h->lde;
      g.C = h->Wl[l].g; g.ldc = h->lde; g.M = l == 0 ? N + 2 : e + 1; g.N = e; g.K = nrows;
      g.epi.kind = EPI_STORE;
      g.b_scale = gsc;
      const bool uid_apart = l == 0 && low_precision(h);
      if (uid_apart) g.M = N + 1;      // the float(uid) row of W_0_ext gets its gradient from the fp32 reduction below
      // TF-Adam runs in the epilogue of the gradient GEMM, as for GANMF's two tensors -- the gradient is never stored, no
      // adam_dense pass.  Layer 0 holds nearly all discriminator parameters ([N+2, e]) and nothing reads W_0 after its gradient
      // in this step; a hidden layer's W_l is read once more, by the backward product dz_{l-1} = dz_l . W_l^T, which therefore
      // goes FIRST (below).  Not with a communicator (the gradient must be reduced first).  In the low-precision modes the
      // float(uid) row of layer 0 is left out of the GEMM and updated by the fp32 kernel below.
      const bool fuse = h->fuse_adam && !h->has_comm && (l == 0 || h->dis_fuse_hidden);
      h->dis_fused[l] = 0;
      if (l > 0 && fuse) TRY(backward(l));
      if (fuse) {
        const bool regD = h->cfg.d_reg != 0.f;
        g.epi.kind = EPI_ADAM; g.epi.adam_theta = h->Wl[l].p; g.epi.adam_m = h->Wl[l].m; g.epi.adam_v = h->Wl[l].v;
        g.epi.adam_alpha = h->scal + S_ALPHA_D; g.epi.adam_reg = h->cfg.d_reg;
        g.epi.sq_partials = regD ? h->dis_slot + (size_t)(4 + l) * h->dis_cap : nullptr;
        GemmTune ft;
        ft.tile = 64; ft.ring = 2; ft.nsplit = 1;
        ft.mode = h->tune.mode != MFMA_AUTO ? h->tune.mode : h->fused_mode;
        ft.bk = h->fused_bk;
        int regn = ADAM_GRID;
        TRY(run_gemm(h, T_DIS_GW, T_RED_DIS_GW, g, true, true, &regn, 24.0 * h->Wl[l].count(), 0, &ft));
        h->dis_fused[l] = 1;
        h->dis_regn[l] = regn;
        if (l > 0) { std::swap(cur, nxt); continue; }      // (the backward product of this layer is done)
      } else
      TRY(run_gemm(h, T_DIS_GW, T_RED_DIS_GW, g, true, true));
      if (uid_apart) {
        const size_t ro = (size_t)(N + 1) * h->lde;      // the float(uid) row of W_0_ext
        const int nblk = (e + 63) / 64;
        const bool apply = h->dis_fused[0] && h->dis_regn[0] + nblk <= h->dis_cap;
        if (h->dis_fused[0] && !apply) return fail(-1, "dis_backprop_hidden: no room for the uid row's sum(theta^2) partials");
        const bool regD = h->cfg.d_reg != 0.f;
        GANMF_LAUNCH(dis_uid_grad_kernel, dim3(nblk), dim3(64 * UIDG_GROUPS), 0, h->st, h->XF, h->ldN, N + 1, cur, h->lde,
                           nrows, e, h->Wl[0].g + ro, apply ? h->Wl[0].p + ro : nullptr, h->Wl[0].m + ro, h->Wl[0].v + ro,
                           h->scal, (int)S_ALPHA_D, h->cfg.d_reg,
                           (apply && regD) ? h->dis_slot + (size_t)4 * h->dis_cap + h->dis_regn[0] : nullptr);
        HIP_TRY(hipGetLastError());
        if (apply && regD) h->dis_regn[0] += nblk;
      }
    }
    if (l > 0) {
      TRY(backward(l));
      std::swap(cur, nxt);
    }
  }
  *dz0_out = cur;
  return 0;
}

int dis_d_step(ganmf_handle* h, const int* rows_dev, int nb, int b_global, float* parts) {
  const int e = h->e;
  const float inv_b = 1.0f / (float)b_global;
  const bool fuse_wo = h->fuse_adam && !h->has_comm && h->dis_fuse_hidden;      // output layer updated by the kernel that forms its gradient
  if (nb > 0) {
    TRY(dis_forward(h, rows_dev, nb, 0));
    float* feat = h->Al[h->L - 1];
    {
      Scope s(h, T_DIS_HEAD, 0, 4.0 * 2 * nb * e * 3);
      GANMF_LAUNCH(dis_head_kernel, dim3((2 * nb + 3) / 4), dim3(256), 0, h->st, feat, h->lde, e + 1,
                         h->Wo.p, 0, 2 * nb, nb, inv_b, h->dlogit, h->dis_slot, h->dis_slot + h->dis_cap);
      GANMF_LAUNCH(dis_dz_top_kernel, dim3(dis_dz_top_blocks(e)), dim3(DZ_COLS * DZ_GROUPS), 0, h->st, feat, h->lde, e,
                         h->Wo.p, h->dlogit, 0, 2 * nb, 0, 0.f, h->act, h->dz0, h->Wo.g, (float*)nullptr,
                         fuse_wo ? h->Wo.p : nullptr, h->Wo.m, h->Wo.v, h->scal, (int)S_ALPHA_D, h->cfg.d_reg,
                         (fuse_wo && h->cfg.d_reg != 0.f) ? h->dis_slot + (size_t)(4 + h->L) * h->dis_cap : nullptr);
      HIP_TRY(hipGetLastError());
    }
    float* dz0;
    TRY(dis_backprop_hidden(h, 0, 2 * nb, true, b_global, &dz0));
  } else {
    GANMF_LAUNCH(open_step_kernel, dim3(1), dim3(64), 0, h->st, h->scal, 0, S_ALPHA_D, h->cfg.d_lr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(h->gD, 0, h->gD_elems * sizeof(float), h->st));
  }
  const bool reg = h->cfg.d_reg != 0.f;
  if (h->has_comm) {
    for (int l = 0; l < h->L; ++l)
      TRY(dp_update(h, T_ADAM_D, h->Wl[l], S_ALPHA_D, h->cfg.d_reg, reg ? h->dis_slot + (size_t)(4 + l) * h->dis_cap : nullptr, 0));
    TRY(dp_update(h, T_ADAM_D, h->Wo, S_ALPHA_D, h->cfg.d_reg, reg ? h->dis_slot + (size_t)(4 + h->L) * h->dis_cap : nullptr, 0));
  } else {
    for (int l = 0; l < h->L; ++l) {
      if (nb > 0 && h->dis_fused[l]) continue;      // updated in the epilogue of its gradient GEMM
      TRY(adam_dense(h, T_ADAM_D, h->Wl[l], h->Wl[l].g, S_ALPHA_D, h->cfg.d_reg, reg ? h->dis_slot + (size_t)(4 + l) * h->dis_cap : nullptr));
    }
    if (!(nb > 0 && fuse_wo))      // (else: updated by dis_dz_top_kernel, which forms its gradient)
      TRY(adam_dense(h, T_ADAM_D, h->Wo, h->Wo.g, S_ALPHA_D, h->cfg.d_reg, reg ? h->dis_slot + (size_t)(4 + h->L) * h->dis_cap : nullptr));
  }
  // parts = {sum sce(real), sum sce(fake), sum theta_D^2}: summed from the slot once per epoch (arenas_finish)
  (void)parts; (void)reg;
  return 0;
}

int dis_g_step(ganmf_handle* h, const int* rows_dev, int nb, int start, int b_global, float* parts) {
  const int N = h->N, e = h->e;
  const float alpha = h->cfg.recon_coefficient;
  const float inv_b = 1.0f / (float)b_global;
  int fmn = 0;
  if (nb > 0) {
    TRY(dis_forward(h, rows_dev, nb, 1));
    float* feat = h->Al[h->L - 1];
    const float fmc = alpha * 2.0f / ((float)b_global * (float)e);
    {  // generator loss = loss_fake + alpha * FM  (DisGANMF.py:135-136): generated rows only, label 0
      Scope s(h, T_DIS_HEAD, 0, 4.0 * 2 * nb * e * 3);
      GANMF_LAUNCH(dis_head_kernel, dim3((nb + 3) / 4), dim3(256), 0, h->st, feat, h->lde, e + 1, h->Wo.p,
                         nb, nb, nb, inv_b, h->dlogit, (float*)nullptr, h->dis_slot);      // generated rows only -> seg 0
      fmn = dis_dz_top_blocks(e);
      GANMF_LAUNCH(dis_dz_top_kernel, dim3(fmn), dim3(DZ_COLS * DZ_GROUPS), 0, h->st, feat, h->lde, e, h->Wo.p, h->dlogit,
                         nb, nb, nb, fmc, h->act, h->dz0, (float*)nullptr, h->dis_slot + h->dis_cap, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                         h->scal, 0, 0.f, (float*)nullptr);
      HIP_TRY(hipGetLastError());
    }
    float* dz0;
    TRY(dis_backprop_hidden(h, nb, nb, false, b_global, &dz0));
    {  // dF = dz_0 . W_0[profile rows]^T     (the uid column of the input is dropped, DisGANMF.py:59)
      GemmP g{};
      g.A = dz0 + (size_t)nb * h->lde; g.lda = h->lde; g.B = h->Wl[0].p; g.ldb = h->lde;
      g.C = h->dF; g.ldc = h->ldN; g.M = nb; g.N = N; g.K = e; g.epi.kind = EPI_STORE;
      g.a_scale = grad_scale(h, b_global);
      TRY(run_gemm(h, T_GEMM_DF, T_RED_DF, g, false, false));
    }
  } else {
    GANMF_LAUNCH(open_step_kernel, dim3(1), dim3(64), 0, h->st, h->scal, 1, S_ALPHA_G, h->cfg.g_lr);
    HIP_TRY(hipGetLastError());
  }
  const bool reg = h->cfg.g_reg != 0.f;
  int regn_v = ADAM_GRID;
  TRY(gen_update(h, nb, start, b_global, &regn_v, h->dis_slot + (size_t)2 * h->dis_cap, h->dis_slot + (size_t)3 * h->dis_cap));
  // parts = {sum sce(fake), sum (feat_f - feat_r)^2, sum U^2, sum V^2}: summed from the slot once per epoch (arenas_finish)
  (void)parts; (void)reg; (void)fmn;
  return 0;
}

// model dispatch
inline size_t arena_stride(const ganmf_handle* h) {      // floats per step: GANMF 4 segments of reg_cap, DisGANMF 5 + L of dis_cap
  return h->cfg.model == GANMF_MODEL_GANMF ? (size_t)4 * h->reg_cap : (size_t)(5 + h->L) * h->dis_cap;
}
int any_d_step(ganmf_handle* h, const int* rows_dev, int nb, int b_global, int64_t idx) {
  float* parts = h->d_parts + 4 * idx;
  float* slot = h->d_arena + (size_t)idx * arena_stride(h);
  if (h->cfg.model == GANMF_MODEL_GANMF) return d_step(h, rows_dev, nb, b_global, parts, slot);
  h->dis_slot = slot;
  return dis_d_step(h, rows_dev, nb, b_global, parts);
}
int any_g_step(ganmf_handle* h, const int* rows_dev, int nb, int start, int b_global, int64_t idx) {
  float* parts = h->g_parts + 4 * idx;
  float* slot = h->g_arena + (size_t)idx * arena_stride(h);
  if (h->cfg.model == GANMF_MODEL_GANMF) return g_step(h, rows_dev, nb, start, b_global, parts, slot);
  h->dis_slot = slot;
  return dis_g_step(h, rows_dev, nb, start, b_global, parts);
}

// zero the per-step arenas before a pass / reduce them into the loss parts after it (GANMF)
int ensure_parts(ganmf_handle* h, int64_t steps);
// Loss parts and per-step arenas of one call, packed into the front of parts_all and zeroed by ONE memset (the call starts on an
// idle queue: every separate fill costs its launch latency; a 20-step call is 2.6 ms of GPU work).
int parts_begin(ganmf_handle* h, int64_t nd, int64_t ng) {
  TRY(ensure_parts(h, std::max<int64_t>(std::max(nd, ng), 1)));
  const size_t stride = arena_stride(h);
  const size_t ndp = (size_t)std::max<int64_t>(nd, 1) * 4, ngp = (size_t)std::max<int64_t>(ng, 1) * 4;
  h->d_parts = h->parts_all;
  h->g_parts = h->d_parts + ndp;
  h->d_arena = h->g_parts + ngp;
  h->g_arena = h->d_arena + (size_t)std::max<int64_t>(nd, 0) * stride;
  const size_t total = ndp + ngp + (size_t)(std::max<int64_t>(nd, 0) + std::max<int64_t>(ng, 0)) * stride;
  HIP_TRY(hipMemsetAsync(h->parts_all, 0, total * sizeof(float), h->st));
  return 0;
}
int arenas_finish(ganmf_handle* h, int64_t nd, int64_t ng) {
  const size_t stride = arena_stride(h);
  Scope s(h, T_MULTIRED, 0, 4.0 * (nd + ng) * stride);
  if (h->cfg.model == GANMF_MODEL_GANMF) {
    if (nd > 0) GANMF_LAUNCH(finish_parts_kernel, dim3((int)nd), dim3(256), 0, h->st, h->d_arena, h->reg_cap, 0, h->d_parts);
    if (ng > 0) GANMF_LAUNCH(finish_parts_kernel, dim3((int)ng), dim3(256), 0, h->st, h->g_arena, h->reg_cap, 1, h->g_parts);
  } else {
    if (nd > 0) GANMF_LAUNCH(finish_dis_parts_kernel, dim3((int)nd), dim3(256), 0, h->st, h->d_arena, (long long)stride, h->dis_cap, 5 + h->L, 0, h->d_parts);
    if (ng > 0) GANMF_LAUNCH(finish_dis_parts_kernel, dim3((int)ng), dim3(256), 0, h->st, h->g_arena, (long long)stride, h->dis_cap, 5 + h->L, 1, h->g_parts);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// pinned host staging (ints: permutation | inverse permutation; floats: loss parts), grown on demand
int ensure_stage(ganmf_handle* h, size_t n_int, size_t n_float) {
  if (n_int > h->stage_i_cap) {
    if (h->stage_i) hipHostFree(h->stage_i);
    h->stage_i = nullptr; h->stage_i_cap = 0;
    HIP_TRY(hipHostMalloc((void**)&h->stage_i, n_int * sizeof(int), hipHostMallocDefault));
    h->stage_i_cap = n_int;
  }
  if (n_float > h->stage_f_cap) {
    if (h->stage_f) hipHostFree(h->stage_f);
    h->stage_f = nullptr; h->stage_f_cap = 0;
    const size_t want = n_float + n_float / 2 + 256;
    HIP_TRY(hipHostMalloc((void**)&h->stage_f, want * sizeof(float), hipHostMallocDefault));
    h->stage_f_cap = want;
  }
  return 0;
}

int ensure_parts(ganmf_handle* h, int64_t steps) {
  if (steps <= h->parts_cap) return 0;
  HIP_TRY(hipStreamSynchronize(h->st));
  // capacity is only published once every buffer of the new size exists: a failing dalloc leaves cap 0 and null
  // pointers, so the next call allocates again instead of running on freed memory
  hipFree(h->parts_all); hipFree(h->colbuf);
  h->parts_all = h->d_parts = h->g_parts = h->d_arena = h->g_arena = h->colbuf = nullptr;
  h->parts_cap = 0;
  const int64_t cap = steps + 64;
  TRY(dalloc(&h->parts_all, (size_t)cap * 2 * (4 + arena_stride(h))));
  TRY(dalloc(&h->colbuf, (size_t)cap));
  h->parts_cap = cap;
  return 0;
}

// losses from the per-step parts (fp32 host arithmetic, same expression order as the oracle)
void finish_losses(const ganmf_handle* h, const std::vector<float>& dp, const std::vector<float>& gp,
                   const std::vector<int>& bglob, int64_t nd, int64_t ng, int64_t per_pass, float* d_losses,
                   float* g_losses) {
  const float alpha = h->cfg.recon_coefficient;
  if (h->cfg.model == GANMF_MODEL_DISGANMF) {
    for (int64_t i = 0; i < nd && d_losses; ++i) {
      const float bg = (float)bglob[i % per_pass];
      const float sq = dp[4 * i + 2];     // (data-parallel: every rank summed its own slice; all-reduced = the whole tensor)
      d_losses[i] = (dp[4 * i] / bg + dp[4 * i + 1] / bg) + h->cfg.d_reg * (sq / 2.0f);
    }
    for (int64_t i = 0; i < ng && g_losses; ++i) {
      const float bg = (float)bglob[i % per_pass];
      const float sv = gp[4 * i + 3];
      g_losses[i] = (gp[4 * i] / bg + alpha * (gp[4 * i + 1] / (bg * (float)h->e))) + h->cfg.g_reg * ((gp[4 * i + 2] + sv) / 2.0f);
    }
    return;
  }
  for (int64_t i = 0; i < nd && d_losses; ++i)
    d_losses[i] = dp[4 * i] + h->cfg.d_reg * (dp[4 * i + 2] / 2.0f);
  for (int64_t i = 0; i < ng && g_losses; ++i) {
    const float bg = (float)bglob[i % per_pass];
    const float Lf = gp[4 * i] / (bg * (float)h->N);
    const float fm = gp[4 * i + 1] / (bg * (float)h->e);
    const float sv = gp[4 * i + 3];       // (data-parallel: slices of V, all-reduced)
    g_losses[i] = ((1.0f - alpha) * Lf + alpha * fm) + h->cfg.g_reg * ((gp[4 * i + 2] + sv) / 2.0f);
  }
}

}  // namespace

// =================================================================================================
extern "C" {

int ganmf_abi_version(void) { return GANMF_ABI_VERSION; }
const char* ganmf_last_error(void) { return g_err.c_str(); }

// CRC-32C, slicing-by-8 (reflected polynomial 0x82F63B78); host only.
static uint32_t g_crc_tab[8][256];
static std::once_flag g_crc_once;
static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xff];
}
uint32_t ganmf_crc32c(uint32_t crc, const void* data, uint64_t n) {
  std::call_once(g_crc_once, crc_init);
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_crc_tab[7][w & 0xff] ^ g_crc_tab[6][(w >> 8) & 0xff] ^ g_crc_tab[5][(w >> 16) & 0xff] ^
        g_crc_tab[4][(w >> 24) & 0xff] ^ g_crc_tab[3][(w >> 32) & 0xff] ^ g_crc_tab[2][(w >> 40) & 0xff] ^
        g_crc_tab[1][(w >> 48) & 0xff] ^ g_crc_tab[0][(w >> 56) & 0xff];
    p += 8; n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xff];
  return ~c;
}

int ganmf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int create_impl(const ganmf_cfg* cfg, ganmf_handle* h);

int ganmf_create(const ganmf_cfg* cfg, ganmf_handle** out) {
  if (!cfg || !out) return fail(-1, "ganmf_create: null argument");
  if (cfg->abi_version != GANMF_ABI_VERSION) return fail(-1, "ganmf_create: ABI version %d != %d", cfg->abi_version, GANMF_ABI_VERSION);
  if (cfg->model != GANMF_MODEL_GANMF && cfg->model != GANMF_MODEL_DISGANMF) return fail(-1, "ganmf_create: unknown model %d", cfg->model);
  if (cfg->model == GANMF_MODEL_DISGANMF && (cfg->d_layers < 1 || cfg->d_layers > 16 || cfg->d_act < 0 || cfg->d_act > 3))
    return fail(-1, "ganmf_create: DisGANMF needs 1 <= d_layers <= 16 and a known activation");
  if (cfg->num_users < 1 || cfg->num_items < 1 || cfg->num_factors < 1 || cfg->emb_dim < 1 || cfg->batch_size < 1)
    return fail(-1, "ganmf_create: non-positive dimension");
  if (cfg->num_users > (1LL << 30) || cfg->num_items > (1LL << 30)) return fail(-1, "ganmf_create: dimension too large");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(-2, "ganmf_create: no HIP device (the HIP path is the only path; there is no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(-1, "ganmf_create: device %d out of range [0,%d)", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));
  ganmf_handle* h = new ganmf_handle();
  const int rc = create_impl(cfg, h);
  if (rc != 0) {
    // a failed create owns nothing afterwards: every buffer / stream / event allocated so far is released (destroy
    // tolerates null members) and a sticky hipErrorOutOfMemory is cleared, so the caller (tune.py maps MemoryError to
    // fitness 0 and keeps running trials in the same process) neither leaks HBM nor sees a spurious error later
    const std::string msg = g_err;
    ganmf_destroy(h);
    (void)hipGetLastError();
    g_err = msg;
    *out = nullptr;
    return rc;
  }
  *out = h;
  return 0;
}

static int create_impl(const ganmf_cfg* cfg, ganmf_handle* h) {
  h->cfg = *cfg;
  h->dev = cfg->device;
  h->U = (int)cfg->num_users; h->N = (int)cfg->num_items; h->k = cfg->num_factors; h->e = cfg->emb_dim;
  h->B = (int)std::min<int64_t>(cfg->batch_size, cfg->num_users);
  h->ldN = round_up(h->N + 2, LD_ALIGN); h->ldk = round_up(h->k + 1, LD_ALIGN); h->lde = round_up(h->e + 1, LD_ALIGN);
  h->tune.tile = env_int("GANMF_TILE", 0);
  if (h->tune.tile != 0 && h->tune.tile != 64 && h->tune.tile != 128) h->tune.tile = 0;
  h->tune.ring = env_int("GANMF_RING", 0);
  if (h->tune.ring != 0 && h->tune.ring != 2 && h->tune.ring != 3 && h->tune.ring != 4) h->tune.ring = 0;
  h->tune.nsplit = std::max(0, env_int("GANMF_NSPLIT", 0));
  h->tune.mode = env_mfma_mode((cfg->flags & GANMF_FLAG_MFMA_F16) ? MFMA_F16 : (cfg->flags & GANMF_FLAG_MFMA_BF16) ? MFMA_BF16 : (cfg->flags & GANMF_FLAG_MFMA_F32) ? MFMA_F32 : MFMA_DEFAULT);
  h->tune.persist = env_int("GANMF_PERSIST", -1);
  h->tune.kg = env_int("GANMF_KG", 0);
  h->tune.tile_order = env_int("GANMF_TILE_ORDER", 0);
  if (h->tune.kg != 0 && h->tune.kg != 1 && h->tune.kg != 2 && h->tune.kg != 4) h->tune.kg = 0;
  h->debug_plan = env_int("GANMF_DEBUG_PLAN", 0) != 0;
  h->fused_mode = env_int("GANMF_FUSED_X3", 1) ? MFMA_BF16X3 : MFMA_F32;
  h->fused_tile = env_int("GANMF_FUSED_TILE", 64) == 128 ? 128 : 64;
  h->fused_bk = env_int("GANMF_FUSED_BK", 32);   // 24 KiB of LDS per workgroup: six co-resident workgroups hide the
                                                 // theta / m / v round trip of each other (33.6 / 28.8 us against 37.2 / 32.0 at 64)
  HIP_TRY(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking));
  // lane events order two streams of THIS device only: no system-scope fence when they are recorded (the cache write-back /
  // invalidate a host-visible event performs showed as ~7 us of idle main lane per fork or join in the data-parallel step's
  // rocprofv3 timeline, profiles/r03_dp_timeline.md; kernel boundaries keep their device-scope release / acquire)
  // With more than one rank the events also order RCCL's peer writes (all-gather results land in We / Wd / V from other GPUs)
  // before the main-lane kernels that read them: the fence-free form has only ever run on one GPU, so world_size > 1 keeps the
  // system-scope fence until a multi-GPU bitwise-replica run has passed without it (GANMF_LANE_EVENT_FENCE = 0 / 1 overrides).
  const bool lane_fence = env_int("GANMF_LANE_EVENT_FENCE", cfg->world_size > 1 ? 1 : 0) != 0;
  const unsigned lane_flags = lane_fence ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
  HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, lane_flags));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_join, lane_flags));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_mid, lane_flags));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_we, lane_flags));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_wd, lane_flags));
  h->fuse_adam = env_int("GANMF_FUSE_ADAM", 1) != 0;
  h->defer_gub = env_int("GANMF_DEFER_GUB", 1) != 0;
  h->multi = env_int("GANMF_MULTI", 31);
  h->dis_fuse_hidden = env_int("GANMF_DIS_FUSE_HIDDEN", 1) != 0;
  h->x3kg = env_int("GANMF_X3KG", 7);      // +10 % steps/s at the ML-1M shape (profiles/r03_gemm_stamps.md)
  h->dcoef_spread = env_int("GANMF_DCOEF_SPREAD", 1) != 0;
  h->pair_ring = env_int("GANMF_PAIR_RING", 2) == 3 ? 3 : 2;
  h->inkernel_reduce = env_int("GANMF_INKERNEL_REDUCE", 0) != 0;   // measured slower than the chip-wide reduce kernel
  h->inlaunch_max = env_int("GANMF_INLAUNCH_MAX", 4);
  h->inlaunch_tags = env_int("GANMF_INLAUNCH_TAGS", 0);
  h->force_coll = env_int("GANMF_FORCE_COLLECTIVES", 0) != 0;
  h->score_presplit = env_int("GANMF_SCORE_PRESPLIT", 1) != 0;
  h->fork_attach = env_int("GANMF_FORK_ATTACH", 1) != 0;
  h->adam_nfast = env_int("GANMF_ADAM_NFAST", 3);
  h->red_elems = env_int("GANMF_RED_ELEMS", 0);
  h->merge_decode = env_int("GANMF_MERGE_DECODE", 1) != 0;
  TRY(dalloc((float**)&h->counters, COUNTER_CAP));
  TRY(dalloc((float**)&h->counters2, COUNTER_CAP));
  const int U = h->U, N = h->N, k = h->k, e = h->e, B = h->B;
  const bool dis = cfg->model == GANMF_MODEL_DISGANMF;
  // parameters; D gradients contiguous for a single all-reduce
  const int world = std::max(1, (int)cfg->world_size);
  TRY(alloc_tensor(h->Ue, U, k, false, 1));      // rows of U belong to their rank: never communicated
  TRY(alloc_tensor(h->V, N, k, true, world));
  TRY(dalloc(&h->V_alt, h->V.cap));
  if (!dis) {
    TRY(alloc_tensor(h->We, N + 1, e, false, world));   // We_ext: row N = encoder bias
    TRY(alloc_tensor(h->Wd, e + 1, N, false, world, h->ldN));   // Wd_ext: row e = decoder bias; shares the leading dimension of the [.., N] work buffers
    h->gD_elems = h->We.cap + h->Wd.cap;
    TRY(dalloc(&h->gD, h->gD_elems));
    h->We.g = h->gD;
    h->Wd.g = h->We.g + h->We.cap;
  } else {
    h->L = cfg->d_layers; h->act = cfg->d_act;
    h->Wl.resize(h->L);
    h->dis_fused.assign(h->L, 0);
    h->dis_regn.assign(h->L, ADAM_GRID);
    h->gD_elems = 0;
    for (int l = 0; l < h->L; ++l) {
      TRY(alloc_tensor(h->Wl[l], l == 0 ? N + 2 : e + 1, e, false, world));
      h->gD_elems += h->Wl[l].cap;
    }
    TRY(alloc_tensor(h->Wo, 1, e + 1, false, world));
    h->gD_elems += h->Wo.cap;
    TRY(dalloc(&h->gD, h->gD_elems));
    float* gp = h->gD;
    for (int l = 0; l < h->L; ++l) { h->Wl[l].g = gp; gp += h->Wl[l].cap; }
    h->Wo.g = gp;
  }
  TRY(dalloc(&h->zero_page, 2048 + 64));   // 8 KiB: one 32-byte line per lane of a workgroup
  TRY(dalloc((float**)&h->perm, (size_t)2 * U));
  h->pos = h->perm + U;
  TRY(dalloc(&h->XF, (size_t)2 * B * h->ldN));
  TRY(dalloc(&h->Ub, (size_t)B * h->ldk));
  TRY(dalloc(&h->dF, (size_t)B * h->ldN));
  TRY(dalloc(&h->gUb, (size_t)B * h->ldk));
  // bias-folding ones columns: XF[:, N] = 1 (and E[:, e] = 1 / a_l[:, e] = 1) for every row; epilogues never store there
  std::vector<float> ones((size_t)2 * B, 1.0f);
  HIP_TRY(hipMemcpy2D(h->XF + N, (size_t)h->ldN * 4, ones.data(), 4, 4, (size_t)2 * B, hipMemcpyHostToDevice));
  if (!dis) {
    TRY(dalloc(&h->E, (size_t)2 * B * h->lde));
    TRY(dalloc(&h->Es, (size_t)2 * B * h->lde));
    TRY(dalloc(&h->Dl, (size_t)2 * B * h->ldN));
    TRY(dalloc(&h->dE, (size_t)2 * B * h->lde));
    HIP_TRY(hipMemcpy2D(h->E + e, (size_t)h->lde * 4, ones.data(), 4, 4, (size_t)2 * B, hipMemcpyHostToDevice));
#ifdef GANMF_PERSIST_DIAG_BUILD
    h->wgrad_stream = env_int("GANMF_WGRAD_STREAM", 0) != 0;
#endif
    if (h->wgrad_stream) {      // three bf16 pieces per activation element (dalloc counts floats: two bf16 each)
      h->ps_N = (long long)2 * B * h->ldN; h->ps_e = (long long)2 * B * h->lde;
      TRY(dalloc((float**)&h->pl_XF, (size_t)(3 * h->ps_N + 1) / 2));
      TRY(dalloc((float**)&h->pl_Dl, (size_t)(3 * h->ps_N + 1) / 2));
      TRY(dalloc((float**)&h->pl_Es, (size_t)(3 * h->ps_e + 1) / 2));
      TRY(dalloc((float**)&h->pl_dE, (size_t)(3 * h->ps_e + 1) / 2));
      TRY(dalloc(&h->wgs_dump, 2048 + 64));
    }
  } else {
    h->Al.resize(h->L, nullptr);
    for (int l = 0; l < h->L; ++l) {
      TRY(dalloc(&h->Al[l], (size_t)2 * B * h->lde));
      HIP_TRY(hipMemcpy2D(h->Al[l] + e, (size_t)h->lde * 4, ones.data(), 4, 4, (size_t)2 * B, hipMemcpyHostToDevice));
    }
    TRY(dalloc(&h->dz0, (size_t)2 * B * h->lde));
    TRY(dalloc(&h->dz1, (size_t)2 * B * h->lde));
    TRY(dalloc(&h->dlogit, (size_t)2 * B));
  }
  TRY(dalloc(&h->rs, (size_t)2 * B));
  TRY(dalloc(&h->scal, S_COUNT));
  TRY(dalloc(&h->sqp, (size_t)2 * std::max(GEMM_RED_GRID, ((B + 63) / 64) * ((N + 63) / 64)) + 16));
  {
    auto t64 = [](int a, int b) { return ((a + 63) / 64) * ((b + 63) / 64); };
    h->reg_cap = std::max({ADAM_GRID, (int)GEMM_RED_GRID, t64(N + 2, e), t64(e + 1, N), t64(N, k), t64(B, N), t64(B, e)});
    h->reg_cap = round_up(h->reg_cap + (e + 63) / 64, 64);      // + the float(uid) row's partials (dis_uid_grad_kernel)
    // DisGANMF: a segment also holds the per-row cross-entropies of a batch half or the feature-matching partials
    h->dis_cap = round_up(std::max({h->reg_cap, B, dis_dz_top_blocks(e) + 1}), 64);
  }
  const float pw[4] = {ADAM_B1, ADAM_B2, ADAM_B1, ADAM_B2};
  HIP_TRY(hipMemcpy(h->scal, pw, sizeof pw, hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

int ganmf_destroy(ganmf_handle* h) {
  if (!h) return 0;
  hipSetDevice(h->dev);
  if (h->st) hipStreamSynchronize(h->st);
  if (h->has_comm && !h->local) ncclCommDestroy(h->comm);
  if (h->local) {
    std::lock_guard<std::mutex> lk(h->local->mu);
    h->local->failed = true;            // a group does not outlive any of its members
    h->local->cv.notify_all();
  }
  free_tensor(h->We, false); free_tensor(h->Wd, false); hipFree(h->zero_page); hipFree(h->Es);
  for (auto& t : h->Wl) free_tensor(t, false);
  free_tensor(h->Wo, false);
  for (float* a : h->Al) hipFree(a);
  hipFree(h->dz0); hipFree(h->dz1); hipFree(h->dlogit);
  free_tensor(h->Ue, false); free_tensor(h->V, true); hipFree(h->V_alt);
  hipFree(h->gD); hipFree(h->indptr); hipFree(h->indices); hipFree(h->data); hipFree(h->perm);
  hipFree(h->csc_colptr); hipFree(h->csc_rowidx); hipFree(h->csc_val); hipFree(h->sp_rows);
  if (h->stage_i) hipHostFree(h->stage_i);
  if (h->stage_f) hipHostFree(h->stage_f);
  hipFree(h->test_indptr); hipFree(h->test_indices); hipFree(h->test_gain); hipFree(h->eval_buf);
  hipFree(h->XF); hipFree(h->Ub); hipFree(h->E); hipFree(h->Dl); hipFree(h->dE); hipFree(h->dF); hipFree(h->gUb);
  hipFree(h->slab); hipFree(h->rs); hipFree(h->scal); hipFree(h->sqp);
  hipFree(h->seen_indptr); hipFree(h->seen_indices); hipFree(h->item_mask); hipFree(h->topk_items); hipFree(h->topk_vals); hipFree(h->sc_ids);
  hipFree(h->colbuf); hipFree(h->parts_all); hipFree(h->sc_rows); hipFree(h->sc_out); hipFree(h->sc_pa); hipFree(h->sc_pb);
  for (auto& r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  if (h->st2) hipStreamSynchronize(h->st2);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->ev_mid) hipEventDestroy(h->ev_mid);
  if (h->ev_t0) hipEventDestroy(h->ev_t0);
  if (h->ev_t1) hipEventDestroy(h->ev_t1);
  if (h->ev_we) hipEventDestroy(h->ev_we);
  if (h->ev_wd) hipEventDestroy(h->ev_wd);
  if (h->st2) hipStreamDestroy(h->st2);
  if (h->st) hipStreamDestroy(h->st);
  hipFree(h->slab2); hipFree(h->counters); hipFree(h->counters2);
  hipFree(h->pl_XF); hipFree(h->pl_Dl); hipFree(h->pl_Es); hipFree(h->pl_dE); hipFree(h->wgs_dump); hipFree(h->wgs_table);
  delete h;
  return 0;
}

int ganmf_comm_unique_id(uint8_t out128[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(out128, &id, 128);
  return 0;
}

int ganmf_comm_init_local(ganmf_handle* h, int32_t group_id) {
  if (!h) return fail(-1, "null handle");
  if (h->has_comm) return fail(-1, "ganmf_comm_init_local: handle already has a communicator");
  const int world = h->cfg.world_size, rank = h->cfg.rank;
  if (world < 1 || world > LOCAL_MAX_WORLD || rank < 0 || rank >= world)
    return fail(-1, "ganmf_comm_init_local: world_size %d / rank %d out of range (max %d)", world, rank, LOCAL_MAX_WORLD);
  std::lock_guard<std::mutex> lk(g_local_mu);
  std::shared_ptr<LocalGroup>& slot = g_local_groups[group_id];
  if (!slot || slot->joined == slot->world || slot->failed) {      // first member of a new group with this id
    slot = std::make_shared<LocalGroup>();
    slot->world = world; slot->dev = h->dev;
  }
  if (slot->world != world || slot->dev != h->dev)
    return fail(-1, "ganmf_comm_init_local: group %d is for world_size %d on device %d", group_id, slot->world, slot->dev);
  ++slot->joined;
  h->local = slot;
  h->has_comm = true;
  return 0;
}

int ganmf_comm_init(ganmf_handle* h, const uint8_t id128[128]) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  NCCL_TRY(ncclCommInitRank(&h->comm, h->cfg.world_size, id, h->cfg.rank));
  h->has_comm = true;
  return 0;
}

int ganmf_set_urm_csr(ganmf_handle* h, const int64_t* indptr, const int32_t* indices, const float* data,
                      int64_t n_rows, int64_t n_cols) {
  if (!h || !indptr || (!indices && indptr[n_rows] > 0)) return fail(-1, "ganmf_set_urm_csr: null argument");
  if (n_rows != h->U || n_cols != h->N) return fail(-1, "ganmf_set_urm_csr: shape %lldx%lld != handle %dx%d", (long long)n_rows, (long long)n_cols, h->U, h->N);
  const int64_t nnz = indptr[n_rows];
  if (indptr[0] != 0 || nnz < 0) return fail(-1, "ganmf_set_urm_csr: bad indptr");
  for (int64_t r = 0; r < n_rows; ++r)
    if (indptr[r + 1] < indptr[r]) return fail(-1, "ganmf_set_urm_csr: indptr not monotone at row %lld", (long long)r);
  for (int64_t j = 0; j < nnz; ++j)
    if (indices[j] < 0 || indices[j] >= n_cols) return fail(-1, "ganmf_set_urm_csr: column index %d out of range at %lld", indices[j], (long long)j);
  HIP_TRY(hipSetDevice(h->dev));
  if (h->indptr) { hipFree(h->indptr); hipFree(h->indices); hipFree(h->data); h->indptr = nullptr; }
  HIP_TRY(hipMalloc((void**)&h->indptr, (n_rows + 1) * sizeof(long long)));
  HIP_TRY(hipMalloc((void**)&h->indices, std::max<int64_t>(nnz, 1) * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&h->data, std::max<int64_t>(nnz, 1) * sizeof(float)));
  HIP_TRY(hipMemcpy(h->indptr, indptr, (n_rows + 1) * sizeof(long long), hipMemcpyHostToDevice));
  if (nnz) {
    HIP_TRY(hipMemcpy(h->indices, indices, nnz * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->data, data, nnz * sizeof(float), hipMemcpyHostToDevice));
  }
  h->nnz = nnz;
  h->has_urm = true;
  // below 0.5 % density (LastFM: 0.22 %) the generator step reads the real rows as CSR (GANMF only;
  // DisGANMF feeds the rows themselves to its discriminator).  GANMF_SPARSE = 0 / 1 overrides.
  const double density = (double)nnz / ((double)n_rows * (double)n_cols);
  const int force = env_int("GANMF_SPARSE", -1);
  h->sparse_g = h->cfg.model == GANMF_MODEL_GANMF && (force >= 0 ? force != 0 : density < 0.005);
  // The discriminator step's sparse path trades 4.B.N.e dense FLOPs (real half of the encode GEMM and of the encoder-gradient
  // GEMM) and the [B, N] row expansion for a CSR row-sum, a CSR lookup per residual element and a CSC walk per gradient element:
  // it pays once those FLOPs are worth more than the few microseconds the lookups add to two epilogues (LastFM at the reference's
  // defaults, B = 32, e = 32: 0.07 GFLOP -- dense; at its tuned B = 1024, e = 398: 28.7 GFLOP of a 146 GFLOP step -- sparse).
  // GANMF_SPARSE_D = 0 / 1 overrides (GANMF_SPARSE = 0 switches both paths off).
  const int force_d = env_int("GANMF_SPARSE_D", -1);
  const double flops_saved = 4.0 * (double)h->B * (double)h->N * (double)h->e;
  h->sparse_d = h->cfg.model == GANMF_MODEL_GANMF && force != 0 &&
                (force_d >= 0 ? force_d != 0 : (density < 0.005 && flops_saved >= 2.0e9));
  hipFree(h->csc_colptr); hipFree(h->csc_rowidx); hipFree(h->csc_val);
  h->csc_colptr = nullptr; h->csc_rowidx = nullptr; h->csc_val = nullptr;
  if (h->sparse_d) {      // counting sort by column; rows ascend inside a column because the CSR rows are walked in order
    std::vector<long long> colptr((size_t)n_cols + 1, 0);
    for (int64_t j = 0; j < nnz; ++j) ++colptr[(size_t)indices[j] + 1];
    for (int64_t c = 0; c < n_cols; ++c) colptr[c + 1] += colptr[c];
    std::vector<long long> fill(colptr.begin(), colptr.end() - 1);
    std::vector<int> rowidx((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> val((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t r = 0; r < n_rows; ++r)
      for (int64_t j = indptr[r]; j < indptr[r + 1]; ++j) {
        const long long at = fill[indices[j]]++;
        rowidx[at] = (int)r; val[at] = data[j];
      }
    HIP_TRY(hipMalloc((void**)&h->csc_colptr, (size_t)(n_cols + 1) * sizeof(long long)));
    HIP_TRY(hipMalloc((void**)&h->csc_rowidx, rowidx.size() * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->csc_val, val.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(h->csc_colptr, colptr.data(), (size_t)(n_cols + 1) * sizeof(long long), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->csc_rowidx, rowidx.data(), rowidx.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->csc_val, val.data(), val.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!h->sp_rows) TRY(dalloc(&h->sp_rows, (size_t)(h->N + CSC_BIAS_PARTS) * h->lde));
  }
  return 0;
}

int ganmf_tensor_shape(ganmf_handle* h, int tensor_id, int64_t* rows, int64_t* cols) {
  View v;
  if (!h || !find_view(h, tensor_id, &v)) return fail(-1, "unknown tensor id %d", tensor_id);
  if (rows) *rows = v.rows;
  if (cols) *cols = v.cols;
  return 0;
}

static int copy_view(ganmf_handle* h, int tensor_id, int slot, float* host, int64_t n, bool to_device, const char* who) {
  View v;
  if (!h || !host || !find_view(h, tensor_id, &v)) return fail(-1, "%s: unknown tensor id %d", who, tensor_id);
  if (n != (int64_t)v.rows * v.cols) return fail(-1, "%s: tensor %d has %lld elements, got %lld", who, tensor_id, (long long)v.rows * v.cols, (long long)n);
  // data-parallel runs keep the Adam moments of a REPLICATED tensor sharded: rank r updates slice r only (dp_update), the other
  // slices of its m / v buffers are stale.  Reading or writing them through one rank would silently give a different optimizer.
  if ((slot == GANMF_SLOT_ADAM_M || slot == GANMF_SLOT_ADAM_V) && h->has_comm && h->cfg.world_size > 1 && tensor_id != GANMF_T_USER_EMB)
    return fail(-1, "%s: the Adam moments of replicated tensor %d are sharded over the %d ranks of a data-parallel run "
                    "(each rank holds its slice only); only user_embeddings' moments are whole on their rank", who, tensor_id, h->cfg.world_size);
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  for (int i = 0; i < v.nseg; ++i) {
    const Seg& sg = v.seg[i];
    float* d = slot_ptr(sg.t, slot);
    if (!d) return fail(-1, "%s: bad slot %d", who, slot);
    d += (size_t)sg.row0 * sg.t->ld + sg.col0;
    float* hp = host + (size_t)sg.host_row0 * v.cols;
    // a [e,1] view stored as one row of e floats: host pitch is the segment's own width
    const size_t hpitch = (size_t)sg.cols * 4, dpitch = (size_t)sg.t->ld * 4;
    if (to_device) HIP_TRY(hipMemcpy2D(d, dpitch, hp, hpitch, (size_t)sg.cols * 4, sg.rows, hipMemcpyHostToDevice));
    else HIP_TRY(hipMemcpy2D(hp, hpitch, d, dpitch, (size_t)sg.cols * 4, sg.rows, hipMemcpyDeviceToHost));
  }
  return 0;
}

int ganmf_set_tensor(ganmf_handle* h, int tensor_id, int slot, const float* host, int64_t n) {
  if (h) ++h->param_version;
  return copy_view(h, tensor_id, slot, const_cast<float*>(host), n, true, "ganmf_set_tensor");
}

int ganmf_get_tensor(ganmf_handle* h, int tensor_id, int slot, float* host, int64_t n) {
  return copy_view(h, tensor_id, slot, host, n, false, "ganmf_get_tensor");
}

int ganmf_get_adam_powers(ganmf_handle* h, float out4[4]) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  HIP_TRY(hipMemcpy(out4, h->scal, 4 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ganmf_set_adam_powers(ganmf_handle* h, const float in4[4]) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  HIP_TRY(hipMemcpy(h->scal, in4, 4 * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

int ganmf_train_epoch(ganmf_handle* h, const int32_t* perm, int64_t n, int32_t d_steps, int32_t g_steps,
                      int64_t n_steps_per_pass, const int32_t* global_batch_rows, float* d_losses,
                      float* g_losses) {
  return ganmf_train_epoch_ragged(h, perm, n, d_steps, g_steps, n_steps_per_pass, global_batch_rows, nullptr, d_losses, g_losses);
}

int ganmf_train_epoch_ragged(ganmf_handle* h, const int32_t* perm, int64_t n, int32_t d_steps, int32_t g_steps,
                             int64_t n_steps_per_pass, const int32_t* global_batch_rows,
                             const int32_t* local_batch_rows, float* d_losses, float* g_losses) {
  if (!h || (!perm && n > 0)) return fail(-1, "ganmf_train_epoch: null argument");
  if (!h->has_urm) return fail(-1, "ganmf_train_epoch: ganmf_set_urm_csr has not been called");
  if (n < 0 || n > h->U) return fail(-1, "ganmf_train_epoch: n=%lld out of range", (long long)n);
  if (d_steps < 0 || g_steps < 0) return fail(-1, "ganmf_train_epoch: negative step count");
  const bool dist = h->has_comm && h->cfg.world_size > 1;
  if (dist && !global_batch_rows) return fail(-1, "ganmf_train_epoch: global_batch_rows required when world_size > 1");
  HIP_TRY(hipSetDevice(h->dev));
  ++h->param_version;
  static const bool time_it = getenv("GANMF_TIME_EPOCH") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tp0).count(); };
  double t_prep = 0, t_first = 0, t_enq = 0, t_sync = 0;
  const int B = h->B;
  const int64_t local_steps = local_batch_rows ? n_steps_per_pass : (n + B - 1) / B;
  const int64_t per_pass = std::max(local_steps, n_steps_per_pass);
  if (per_pass == 0) return 0;
  // slice i = rows [slice_at[i], slice_at[i] + slice_nb[i]) of the permutation
  std::vector<int64_t> slice_at(per_pass);
  std::vector<int> slice_nb(per_pass);
  {
    int64_t at = 0;
    for (int64_t i = 0; i < per_pass; ++i) {
      int nb;
      if (local_batch_rows) {
        nb = local_batch_rows[i];
        if (nb < 0 || nb > B || at + nb > n) return fail(-1, "ganmf_train_epoch_ragged: local_batch_rows[%lld]=%d (batch_size %d, %lld of %lld rows used)", (long long)i, nb, B, (long long)at, (long long)n);
      } else {
        nb = (int)std::max<int64_t>(0, std::min<int64_t>(B, n - i * B));
        at = std::min<int64_t>(i * B, n);
      }
      slice_at[i] = at; slice_nb[i] = nb;
      at += nb;
    }
    if (local_batch_rows && at != n) return fail(-1, "ganmf_train_epoch_ragged: local_batch_rows sum to %lld, n = %lld", (long long)at, (long long)n);
  }
  // the permutation and its inverse go up from ONE pinned staging buffer in one asynchronous copy, the loss parts come back
  // into pinned memory: pageable copies are staged synchronously by the runtime, ~0.1 ms per epoch call that a 10-slice call
  // (2.6 ms of GPU work) notices
  TRY(ensure_stage(h, (size_t)2 * h->U, 0));
  int* const stage_perm = h->stage_i;
  int* const pos = h->stage_i + h->U;
  std::fill(pos, pos + h->U, -1);
  for (int64_t i = 0; i < n; ++i) {
    const int r = perm[i];
    if (r < 0 || r >= h->U) return fail(-1, "ganmf_train_epoch: row id %d out of range at %lld", r, (long long)i);
    if (pos[r] != -1) return fail(-1, "ganmf_train_epoch: row id %d appears twice in the permutation", r);
    pos[r] = (int)i;
    stage_perm[i] = r;
  }
  std::vector<int> bglob(per_pass);
  for (int64_t i = 0; i < per_pass; ++i) {
    const int nb = slice_nb[i];
    bglob[i] = global_batch_rows ? global_batch_rows[i] : nb;
    if (bglob[i] < nb || bglob[i] < 1) return fail(-1, "ganmf_train_epoch: global_batch_rows[%lld]=%d < local %d", (long long)i, bglob[i], nb);
  }
  t_prep = since();
  HIP_TRY(hipMemcpyAsync(h->perm, stage_perm, (size_t)2 * h->U * sizeof(int), hipMemcpyHostToDevice, h->st));   // perm | pos, contiguous on both sides
  const int64_t nd = (int64_t)d_steps * per_pass, ng = (int64_t)g_steps * per_pass;
  TRY(parts_begin(h, nd, ng));
  t_first = since();
  int64_t idx = 0;
  for (int p = 0; p < d_steps; ++p)
    for (int64_t i = 0; i < per_pass; ++i, ++idx) {
      const int64_t a = slice_at[i];
      const int nb = slice_nb[i];
      TRY(any_d_step(h, h->perm + std::min<int64_t>(a, std::max<int64_t>(n - 1, 0)), nb, bglob[i], idx));
    }
  idx = 0;
  for (int p = 0; p < g_steps; ++p)
    for (int64_t i = 0; i < per_pass; ++i, ++idx) {
      const int64_t a = slice_at[i];
      const int nb = slice_nb[i];
      TRY(any_g_step(h, h->perm + std::min<int64_t>(a, std::max<int64_t>(n - 1, 0)), nb, (int)a, bglob[i], idx));
    }
  TRY(dp_join(h));      // the last step's side-lane updates (and their sum(theta^2) partials) are complete
  TRY(arenas_finish(h, nd, ng));
  // loss parts are sums over this rank's rows / this rank's slices of the replicated tensors: one all-reduce per epoch.
  // GANMF's D loss itself (column 0) is formed on the device from sums that were all-reduced before the hinge and is
  // global already: only its sum(theta_D^2) column travels.
  if (dist && nd > 0) {
    if (h->cfg.model == GANMF_MODEL_DISGANMF) TRY(allreduce(h, h->d_parts, (size_t)nd * 4));
    else {
      const int grid = (int)std::min<int64_t>(256, (nd + 255) / 256);
      GANMF_LAUNCH(col_copy_kernel, dim3(grid), dim3(256), 0, h->st, h->d_parts, h->colbuf, (long long)nd, 2, 1);
      HIP_TRY(hipGetLastError());
      TRY(allreduce(h, h->colbuf, (size_t)nd));
      GANMF_LAUNCH(col_copy_kernel, dim3(grid), dim3(256), 0, h->st, h->d_parts, h->colbuf, (long long)nd, 2, 0);
      HIP_TRY(hipGetLastError());
    }
  }
  if (dist && ng > 0) TRY(allreduce(h, h->g_parts, (size_t)ng * 4));
  const size_t ndp = (size_t)std::max<int64_t>(nd, 1) * 4, ngp = (size_t)std::max<int64_t>(ng, 1) * 4;
  TRY(ensure_stage(h, 0, ndp + ngp));
  HIP_TRY(hipMemcpyAsync(h->stage_f, h->d_parts, (ndp + ngp) * sizeof(float), hipMemcpyDeviceToHost, h->st));   // d_parts | g_parts, contiguous
  t_enq = since();
  HIP_TRY(hipStreamSynchronize(h->st));
  t_sync = since();
  const std::vector<float> dp(h->stage_f, h->stage_f + ndp), gp(h->stage_f + ndp, h->stage_f + ndp + ngp);
  finish_losses(h, dp, gp, bglob, nd, ng, per_pass, d_losses, g_losses);
  if (time_it)
    fprintf(stderr, "[ganmf epoch] %lld D + %lld G steps: host prep %.0f us, first step enqueued at %.0f, all enqueued at %.0f, synced at %.0f, done at %.0f us\n",
            (long long)nd, (long long)ng, t_prep, t_first, t_enq, t_sync, since());
  return 0;
}

int ganmf_train_step(ganmf_handle* h, int kind, const int32_t* uids, int32_t n, float* loss) {
  if (!h || !uids) return fail(-1, "ganmf_train_step: null argument");
  if (!h->has_urm) return fail(-1, "ganmf_train_step: ganmf_set_urm_csr has not been called");
  if (n < 1 || n > h->B) return fail(-1, "ganmf_train_step: batch of %d rows (handle batch_size %d)", n, h->B);
  if (kind != 0 && kind != 1) return fail(-1, "ganmf_train_step: kind must be 0 (D) or 1 (G)");
  if (h->has_comm && h->cfg.world_size > 1) return fail(-1, "ganmf_train_step: single-GPU entry; use ganmf_train_epoch");
  HIP_TRY(hipSetDevice(h->dev));
  ++h->param_version;
  std::vector<int> pos(h->U, -1);
  for (int i = 0; i < n; ++i) {
    if (uids[i] < 0 || uids[i] >= h->U) return fail(-1, "ganmf_train_step: row id %d out of range", uids[i]);
    if (pos[uids[i]] != -1) return fail(-1, "ganmf_train_step: duplicate row id %d", uids[i]);
    pos[uids[i]] = i;
  }
  HIP_TRY(hipMemcpyAsync(h->perm, uids, n * sizeof(int), hipMemcpyHostToDevice, h->st));
  HIP_TRY(hipMemcpyAsync(h->pos, pos.data(), (size_t)h->U * sizeof(int), hipMemcpyHostToDevice, h->st));
  TRY(parts_begin(h, kind == 0, kind == 1));
  if (kind == 0) TRY(any_d_step(h, h->perm, n, n, 0));
  else TRY(any_g_step(h, h->perm, n, 0, n, 0));
  TRY(dp_join(h));
  TRY(arenas_finish(h, kind == 0, kind == 1));
  std::vector<float> dp(4), gp(4);
  HIP_TRY(hipMemcpyAsync(dp.data(), h->d_parts, 4 * sizeof(float), hipMemcpyDeviceToHost, h->st));
  HIP_TRY(hipMemcpyAsync(gp.data(), h->g_parts, 4 * sizeof(float), hipMemcpyDeviceToHost, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  std::vector<int> bglob(1, n);
  float dl = 0.f, gl = 0.f;
  finish_losses(h, dp, gp, bglob, kind == 0 ? 1 : 0, kind == 1 ? 1 : 0, 1, &dl, &gl);
  if (loss) *loss = kind == 0 ? dl : gl;
  return 0;
}

// The scoring product itself: out[n, ldw] = rows[ids] . cols^T.  Many-tile shapes under the fp32-accurate default arithmetic take
// the pre-split persistent kernel (gemm_bf16p.hpp): both factors are split ONCE into their three bf16 planes (the gather of the
// scored rows rides in that pass), then one persistent launch; everything else goes through the planner (run_gemm).
static int score_product(ganmf_handle* h, const int* ids_dev, int64_t n, int transposed, int W, int ldw, bool gemm_only = false) {
  Tensor& rowsT = transposed ? h->V : h->Ue;
  Tensor& colsT = transposed ? h->Ue : h->V;
  const bool presplit = h->score_presplit && (h->tune.mode == MFMA_AUTO || h->tune.mode == MFMA_BF16X3) &&
                        h->tune.tile != 64 && bf16p_eligible((int)n, W, h->k);
  if (presplit) {
    const int mpad = round_up((int)n, BF16P_TILE), npad = round_up(W, BF16P_TILE), kp2 = round_up(h->k, BF16P_BK) / 2;
    const size_t need_a = (size_t)3 * mpad * kp2, need_b = (size_t)3 * npad * kp2;
    if (need_a > h->sc_pa_cap) {
      HIP_TRY(hipStreamSynchronize(h->st));
      hipFree(h->sc_pa); h->sc_pa = nullptr; h->sc_pa_cap = 0;
      HIP_TRY(hipMalloc((void**)&h->sc_pa, need_a * sizeof(unsigned))); h->sc_pa_cap = need_a;
    }
    if (need_b > h->sc_pb_cap) {
      HIP_TRY(hipStreamSynchronize(h->st));
      hipFree(h->sc_pb); h->sc_pb = nullptr; h->sc_pb_cap = 0;
      HIP_TRY(hipMalloc((void**)&h->sc_pb, need_b * sizeof(unsigned))); h->sc_pb_cap = need_b;
    }
    const double fl = gemm_flops((double)n, W, h->k);
    Scope s(h, T_SCORE_GEMM, fl, gemm_bytes((double)n, W, h->k));
    if (gemm_only) {      // (ganmf_bench_scores: operands prepared by the call before)
      HIP_TRY(gemm_bf16p_launch(h->st, h->sc_pa, mpad, h->sc_pb, npad, kp2, h->sc_out, ldw, (int)n, W));
      return 0;
    }
    // one split pass: the scored rows (gathered through ids) and, unless its planes are still those of the current parameters,
    // the other factor (h->param_version counts every call that can change a parameter)
    const bool b_cached = h->sc_pb_src == colsT.p && h->sc_pb_version == h->param_version && h->sc_pb_rows == W;
    const int ga = (int)std::min<long long>(4096, ((long long)mpad * kp2 + 255) / 256);
    const int gb = b_cached ? 0 : (int)std::min<long long>(4096, ((long long)npad * kp2 + 255) / 256);
    const PresplitJob ja{rowsT.p, h->ldk, ids_dev, (int)n, mpad, h->sc_pa}, jb{colsT.p, h->ldk, nullptr, W, npad, h->sc_pb};
    GANMF_LAUNCH(presplit_rows_kernel, dim3(ga + gb), dim3(256), 0, h->st, ja, jb, ga, h->k, kp2);
    HIP_TRY(hipGetLastError());
    h->sc_pb_src = colsT.p; h->sc_pb_version = h->param_version; h->sc_pb_rows = W;
    HIP_TRY(gemm_bf16p_launch(h->st, h->sc_pa, mpad, h->sc_pb, npad, kp2, h->sc_out, ldw, (int)n, W));
    return 0;
  }
  const size_t need_rows = (size_t)n * h->ldk;
  if (need_rows > h->sc_rows_cap) {
    HIP_TRY(hipStreamSynchronize(h->st));
    hipFree(h->sc_rows); h->sc_rows = nullptr; h->sc_rows_cap = 0;
    TRY(dalloc(&h->sc_rows, need_rows)); h->sc_rows_cap = need_rows;
  }
  if (!gemm_only) {
    const long long total = (long long)n * (h->ldk / 4);
    GANMF_LAUNCH(gather_rows_kernel, dim3((int)std::min<long long>(2048, (total + 255) / 256)), dim3(256), 0,
                       h->st, rowsT.p, h->ldk, ids_dev, (int)n, h->sc_rows);
    HIP_TRY(hipGetLastError());
  }
  GemmP g{};
  g.A = h->sc_rows; g.lda = h->ldk; g.B = colsT.p; g.ldb = h->ldk;
  g.C = h->sc_out; g.ldc = ldw; g.M = (int)n; g.N = W; g.K = h->k; g.epi.kind = EPI_STORE; g.c_pad_writable = 1;
  return run_gemm(h, T_SCORE_GEMM, T_RED_SCORE, g, false, false);
}

static int scores_device(ganmf_handle* h, const int* ids_dev, int64_t n, int transposed, float** out_dev, int* width,
                         int* ld_out) {
  Tensor& colsT = transposed ? h->Ue : h->V;   // the other factor
  const int W = colsT.rows, ldw = round_up(W, LD_ALIGN);
  const size_t need_out = (size_t)n * ldw;
  if (need_out > h->sc_out_cap) {
    HIP_TRY(hipStreamSynchronize(h->st));
    hipFree(h->sc_out); h->sc_out = nullptr; h->sc_out_cap = 0;
    TRY(dalloc(&h->sc_out, need_out)); h->sc_out_cap = need_out;
  }
  TRY(score_product(h, ids_dev, n, transposed, W, ldw));
  *out_dev = h->sc_out; *width = W; *ld_out = ldw;
  return 0;
}

// device copy of an id list in the handle's reusable buffer (scores / recommend are called once per 1000-user
// block by the evaluators: no allocation per call)
static int upload_ids(ganmf_handle* h, const int32_t* ids, int64_t n, int** out) {
  if ((size_t)n > h->sc_ids_cap) {
    HIP_TRY(hipStreamSynchronize(h->st));
    if (h->sc_ids) hipFree(h->sc_ids);
    h->sc_ids = nullptr; h->sc_ids_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->sc_ids, (size_t)n * sizeof(int)));
    h->sc_ids_cap = (size_t)n;
  }
  HIP_TRY(hipMemcpyAsync(h->sc_ids, ids, n * sizeof(int), hipMemcpyHostToDevice, h->st));
  *out = h->sc_ids;
  return 0;
}

// The filter of ganmf_set_score_filter as the two kernel arguments (item mask, indptr of the rows whose emptiness means "cold"),
// checked against the orientation in use: score width W, `limit` scored rows.
static int score_filter_args(ganmf_handle* h, const char* who, int W, int limit, const unsigned char** mask, const long long** cold) {
  *mask = nullptr; *cold = nullptr;
  if (h->item_mask_w > 0) {
    if (h->item_mask_w > W) return fail(-1, "%s: the score filter lists item %lld but the score rows have %d columns", who, (long long)h->item_mask_w - 1, W);
    *mask = h->item_mask;
  }
  if (h->mask_cold) {
    if (!h->seen_indptr || h->seen_rows != limit || h->seen_cols != W)
      return fail(-1, "%s: masking cold rows needs ganmf_set_seen_csr with a %d x %d matrix", who, limit, W);
    *cold = h->seen_indptr;
  }
  return 0;
}
static int apply_score_filter(ganmf_handle* h, const char* who, float* scores, int ld, int W, const int* ids_dev, int64_t n, int limit) {
  const unsigned char* mask; const long long* cold;
  TRY(score_filter_args(h, who, W, limit, &mask, &cold));
  if (!mask && !cold) return 0;
  GANMF_LAUNCH(score_filter_kernel, dim3((int)n), dim3(256), 0, h->st, scores, ld, W, ids_dev, mask, cold);
  HIP_TRY(hipGetLastError());
  return 0;
}

int ganmf_set_score_filter(ganmf_handle* h, const int32_t* items, int64_t n_items, int mask_cold_rows) {
  if (!h) return fail(-1, "null handle");
  if (n_items < 0 || (n_items > 0 && !items)) return fail(-1, "ganmf_set_score_filter: bad item list");
  const int64_t wmax = std::max(h->U, h->N);
  int64_t top = 0;
  for (int64_t i = 0; i < n_items; ++i) {
    if (items[i] < 0 || items[i] >= wmax) return fail(-1, "ganmf_set_score_filter: item %d out of range [0,%lld)", items[i], (long long)wmax);
    top = std::max<int64_t>(top, (int64_t)items[i] + 1);
  }
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  if (n_items > 0) {
    if ((size_t)wmax > h->item_mask_cap) {
      hipFree(h->item_mask); h->item_mask = nullptr; h->item_mask_cap = 0;
      HIP_TRY(hipMalloc((void**)&h->item_mask, (size_t)wmax));
      h->item_mask_cap = (size_t)wmax;
    }
    std::vector<unsigned char> m((size_t)wmax, 0);
    for (int64_t i = 0; i < n_items; ++i) m[(size_t)items[i]] = 1;
    HIP_TRY(hipMemcpy(h->item_mask, m.data(), m.size(), hipMemcpyHostToDevice));
  }
  h->item_mask_w = n_items > 0 ? top : 0;
  h->mask_cold = mask_cold_rows != 0;
  return 0;
}

int ganmf_scores(ganmf_handle* h, const int32_t* ids, int64_t n, int transposed, float* out) {
  if (!h || !ids || !out) return fail(-1, "ganmf_scores: null argument");
  if (n < 1 || n > (1 << 30)) return fail(-1, "ganmf_scores: n out of range");
  const int limit = transposed ? h->N : h->U;
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= limit) return fail(-1, "ganmf_scores: id %d out of range [0,%d)", ids[i], limit);
  HIP_TRY(hipSetDevice(h->dev));
  int* ids_dev = nullptr;
  TRY(upload_ids(h, ids, n, &ids_dev));
  float* od = nullptr; int W = 0, ldw = 0;
  int rc = scores_device(h, ids_dev, n, transposed, &od, &W, &ldw);
  if (rc == 0) rc = apply_score_filter(h, "ganmf_scores", od, ldw, W, ids_dev, n, limit);
  if (rc == 0) {
    hipError_t e = hipMemcpy2DAsync(out, (size_t)W * 4, od, (size_t)ldw * 4, (size_t)W * 4, n, hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) rc = fail(-2, "ganmf_scores: copy back failed: %s", hipGetErrorString(e));
  }
  hipStreamSynchronize(h->st);
  return rc;
}

int ganmf_set_seen_csr(ganmf_handle* h, const int64_t* indptr, const int32_t* indices, int64_t n_rows, int64_t n_cols) {
  if (!h || !indptr) return fail(-1, "ganmf_set_seen_csr: null argument");
  const int64_t nnz = indptr[n_rows];
  if (indptr[0] != 0 || nnz < 0 || (nnz > 0 && !indices)) return fail(-1, "ganmf_set_seen_csr: bad indptr");
  for (int64_t r = 0; r < n_rows; ++r)
    if (indptr[r + 1] < indptr[r]) return fail(-1, "ganmf_set_seen_csr: indptr not monotone at row %lld", (long long)r);
  for (int64_t j = 0; j < nnz; ++j)
    if (indices[j] < 0 || indices[j] >= n_cols) return fail(-1, "ganmf_set_seen_csr: column index %d out of range", indices[j]);
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  if (h->seen_indptr) { hipFree(h->seen_indptr); hipFree(h->seen_indices); h->seen_indptr = nullptr; h->seen_indices = nullptr; }
  HIP_TRY(hipMalloc((void**)&h->seen_indptr, (n_rows + 1) * sizeof(long long)));
  HIP_TRY(hipMalloc((void**)&h->seen_indices, std::max<int64_t>(nnz, 1) * sizeof(int)));
  HIP_TRY(hipMemcpy(h->seen_indptr, indptr, (n_rows + 1) * sizeof(long long), hipMemcpyHostToDevice));
  if (nnz) HIP_TRY(hipMemcpy(h->seen_indices, indices, nnz * sizeof(int), hipMemcpyHostToDevice));
  h->seen_rows = n_rows; h->seen_cols = n_cols;
  return 0;
}

// scores -> seen mask -> top-`cutoff` of the rows `ids`, left on the device in h->topk_items / h->topk_vals ([n, cutoff]);
// *ids_dev_out = the uploaded ids.  Shared by ganmf_recommend and ganmf_evaluate.
static int recommend_device(ganmf_handle* h, const char* who, const int32_t* ids, int64_t n, int transposed, int32_t cutoff,
                            int remove_seen, int** ids_dev_out) {
  if (n < 1 || n > (1 << 30)) return fail(-1, "%s: n out of range", who);
  const int limit = transposed ? h->N : h->U, W = transposed ? h->U : h->N;
  if (cutoff < 1 || cutoff > W || cutoff > GANMF_RECOMMEND_MAX_CUTOFF)
    return fail(-1, "%s: cutoff %d out of range [1,%d]", who, cutoff, std::min(W, GANMF_RECOMMEND_MAX_CUTOFF));
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= limit) return fail(-1, "%s: id %d out of range [0,%d)", who, ids[i], limit);
  if (remove_seen && (!h->seen_indptr || h->seen_rows != limit || h->seen_cols != W))
    return fail(-1, "%s: remove_seen needs ganmf_set_seen_csr with a %d x %d matrix", who, limit, W);
  HIP_TRY(hipSetDevice(h->dev));
  int* ids_dev = nullptr;
  TRY(upload_ids(h, ids, n, &ids_dev));
  const size_t need = (size_t)n * cutoff;
  if (need > h->topk_cap) {
    HIP_TRY(hipStreamSynchronize(h->st));
    hipFree(h->topk_items); hipFree(h->topk_vals);
    h->topk_items = nullptr; h->topk_vals = nullptr; h->topk_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->topk_items, need * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->topk_vals, need * sizeof(float)));
    h->topk_cap = need;
  }
  const unsigned char* fmask; const long long* fcold;
  TRY(score_filter_args(h, who, W, limit, &fmask, &fcold));
  float* od = nullptr; int Wd = 0, ldw = 0;
  TRY(scores_device(h, ids_dev, n, transposed, &od, &Wd, &ldw));
  const int lds_cap = 32768;   // floats: 128 KiB of the CU's 160 KiB
  const size_t shmem = Wd <= lds_cap ? (size_t)Wd * sizeof(float) : 0;
  if (shmem > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mask_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  GANMF_LAUNCH(mask_topk_kernel, dim3((int)n), dim3(256), shmem, h->st, od, ldw, Wd, ids_dev,
                     remove_seen ? h->seen_indptr : (const long long*)nullptr, h->seen_indices, (int)cutoff, lds_cap,
                     h->topk_items, h->topk_vals, fmask, fcold);
  HIP_TRY(hipGetLastError());
  if (ids_dev_out) *ids_dev_out = ids_dev;
  return 0;
}

int ganmf_set_test_csr(ganmf_handle* h, const int64_t* indptr, const int32_t* indices, const double* gains, int64_t n_rows,
                       int64_t n_cols) {
  if (!h || !indptr || (!indices && indptr[n_rows] > 0) || (!gains && indptr[n_rows] > 0)) return fail(-1, "ganmf_set_test_csr: null argument");
  if (n_rows < 1 || n_cols < 1) return fail(-1, "ganmf_set_test_csr: empty matrix");
  const int64_t nnz = indptr[n_rows];
  for (int64_t r = 0; r < n_rows; ++r) {
    if (indptr[r + 1] < indptr[r]) return fail(-1, "ganmf_set_test_csr: indptr not monotone at row %lld", (long long)r);
    for (int64_t j = indptr[r]; j < indptr[r + 1]; ++j) {
      if (indices[j] < 0 || indices[j] >= n_cols) return fail(-1, "ganmf_set_test_csr: column %d out of range in row %lld", indices[j], (long long)r);
      if (j > indptr[r] && indices[j] <= indices[j - 1]) return fail(-1, "ganmf_set_test_csr: row %lld is not sorted / has duplicates", (long long)r);
    }
  }
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  hipFree(h->test_indptr); hipFree(h->test_indices); hipFree(h->test_gain);
  h->test_indptr = nullptr; h->test_indices = nullptr; h->test_gain = nullptr; h->test_rows = h->test_cols = 0;
  HIP_TRY(hipMalloc((void**)&h->test_indptr, (size_t)(n_rows + 1) * sizeof(long long)));
  HIP_TRY(hipMalloc((void**)&h->test_indices, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int)));
  HIP_TRY(hipMalloc((void**)&h->test_gain, (size_t)std::max<int64_t>(nnz, 1) * sizeof(double)));
  static_assert(sizeof(long long) == sizeof(int64_t), "indptr width");
  HIP_TRY(hipMemcpy(h->test_indptr, indptr, (size_t)(n_rows + 1) * sizeof(long long), hipMemcpyHostToDevice));
  if (nnz > 0) {
    HIP_TRY(hipMemcpy(h->test_indices, indices, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->test_gain, gains, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
  }
  h->test_rows = n_rows; h->test_cols = n_cols;
  return 0;
}

int ganmf_evaluate(ganmf_handle* h, const int32_t* ids, int64_t n, int transposed, int remove_seen, const int32_t* cutoffs,
                   int32_t n_cutoffs, const double* disc, const double* ideal_cum, double* sums) {
  if (!h || !ids || !cutoffs || !disc || !ideal_cum || !sums) return fail(-1, "ganmf_evaluate: null argument");
  if (n_cutoffs < 1 || n_cutoffs > GANMF_EVAL_MAX_CUTOFFS) return fail(-1, "ganmf_evaluate: 1..%d cut-offs per call", GANMF_EVAL_MAX_CUTOFFS);
  const int limit = transposed ? h->N : h->U, W = transposed ? h->U : h->N;
  if (!h->test_indptr || h->test_rows != limit || h->test_cols != W)
    return fail(-1, "ganmf_evaluate: needs ganmf_set_test_csr with a %d x %d matrix", limit, W);
  int K = 0;
  for (int i = 0; i < n_cutoffs; ++i) {
    if (cutoffs[i] < 1) return fail(-1, "ganmf_evaluate: cut-off %d", cutoffs[i]);
    K = std::max(K, (int)cutoffs[i]);
  }
  int* ids_dev = nullptr;
  TRY(recommend_device(h, "ganmf_evaluate", ids, n, transposed, K, remove_seen, &ids_dev));
  const int grid = (int)((n + 255) / 256);
  const size_t n_part = (size_t)grid * n_cutoffs * EVAL_METRICS;
  const size_t need = (size_t)K + (size_t)n * K + n_part;
  if (need > h->eval_cap) {
    HIP_TRY(hipStreamSynchronize(h->st));
    hipFree(h->eval_buf); h->eval_buf = nullptr; h->eval_cap = 0;
    HIP_TRY(hipMalloc((void**)&h->eval_buf, need * sizeof(double)));
    h->eval_cap = need;
  }
  double* d_disc = h->eval_buf;
  double* d_ideal = d_disc + K;
  double* d_part = d_ideal + (size_t)n * K;
  HIP_TRY(hipMemcpyAsync(d_disc, disc, (size_t)K * sizeof(double), hipMemcpyHostToDevice, h->st));
  HIP_TRY(hipMemcpyAsync(d_ideal, ideal_cum, (size_t)n * K * sizeof(double), hipMemcpyHostToDevice, h->st));
  EvalP p{};
  p.items = h->topk_items; p.K = K; p.n = (int)n; p.ids = ids_dev;
  p.t_indptr = h->test_indptr; p.t_indices = h->test_indices; p.t_gain = h->test_gain;
  p.disc = d_disc; p.ideal_cum = d_ideal; p.ncut = n_cutoffs; p.partials = d_part;
  for (int i = 0; i < n_cutoffs; ++i) p.cutoffs[i] = cutoffs[i];
  GANMF_LAUNCH(eval_topk_kernel, dim3(grid), dim3(256), 0, h->st, p);
  HIP_TRY(hipGetLastError());
  std::vector<double> part(n_part);
  HIP_TRY(hipMemcpyAsync(part.data(), d_part, n_part * sizeof(double), hipMemcpyDeviceToHost, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  for (int i = 0; i < n_cutoffs * EVAL_METRICS; ++i) sums[i] = 0.0;
  for (int b = 0; b < grid; ++b)      // block order: reproducible
    for (int i = 0; i < n_cutoffs * EVAL_METRICS; ++i) sums[i] += part[(size_t)b * n_cutoffs * EVAL_METRICS + i];
  return 0;
}

int ganmf_recommend(ganmf_handle* h, const int32_t* ids, int64_t n, int transposed, int32_t cutoff, int remove_seen,
                    int32_t* out_items, float* out_scores) {
  if (!h || !ids || !out_items) return fail(-1, "ganmf_recommend: null argument");
  int rc = recommend_device(h, "ganmf_recommend", ids, n, transposed, cutoff, remove_seen, nullptr);
  if (rc == 0) {
    const size_t need = (size_t)n * cutoff;
    hipError_t e = hipMemcpyAsync(out_items, h->topk_items, need * sizeof(int), hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess && out_scores) e = hipMemcpyAsync(out_scores, h->topk_vals, need * sizeof(float), hipMemcpyDeviceToHost, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) rc = fail(-2, "ganmf_recommend: %s", hipGetErrorString(e));
  }
  hipStreamSynchronize(h->st);
  return rc;
}

int ganmf_bench_scores(ganmf_handle* h, int64_t n, int transposed, int32_t iters, float* ms_per_launch) {
  if (!h || iters < 1) return fail(-1, "ganmf_bench_scores: bad argument");
  const int limit = transposed ? h->N : h->U;
  if (n < 1 || n > limit) return fail(-1, "ganmf_bench_scores: n out of range");
  HIP_TRY(hipSetDevice(h->dev));
  std::vector<int> ids(n);
  for (int64_t i = 0; i < n; ++i) ids[i] = (int)i;
  int* ids_dev = nullptr;
  TRY(upload_ids(h, ids.data(), n, &ids_dev));
  HIP_TRY(hipStreamSynchronize(h->st));      // `ids` is a local: the copy must be done before it goes away
  float* od; int W, ldw;
  int rc = scores_device(h, ids_dev, n, transposed, &od, &W, &ldw);  // warm-up + allocation
  if (rc) return rc;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const bool was = h->prof;
  h->prof = false;
  // Default: the GEMM launch alone on prepared operands (what this entry has always timed).  GANMF_BENCH_SCORES_PRODUCT=1: the whole
  // product per iteration -- the gather of the scored rows, or on the pre-split kernel BOTH split passes (as the first scoring
  // call after a training epoch pays them), + the GEMM.
  const bool whole = env_int("GANMF_BENCH_SCORES_PRODUCT", 0) != 0;
  score_product(h, ids_dev, n, transposed, W, ldw);
  hipEventRecord(a, h->st);
  for (int i = 0; i < iters; ++i) {
    if (whole) h->sc_pb_version = -1;
    score_product(h, ids_dev, n, transposed, W, ldw, !whole);
  }
  h->prof = was;
  hipEventRecord(b, h->st);
  hipError_t e = hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  if (e != hipSuccess) return fail(-2, "ganmf_bench_scores: %s", hipGetErrorString(e));
  if (ms_per_launch) *ms_per_launch = ms / iters;
  return 0;
}

int ganmf_snapshot_best(ganmf_handle* h) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  for (Tensor* t : all_tensors(h))
    HIP_TRY(hipMemcpyAsync(t->best, t->p, t->padded() * sizeof(float), hipMemcpyDeviceToDevice, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  return 0;
}

int ganmf_restore_best(ganmf_handle* h) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  ++h->param_version;
  for (Tensor* t : all_tensors(h))
    HIP_TRY(hipMemcpyAsync(t->p, t->best, t->padded() * sizeof(float), hipMemcpyDeviceToDevice, h->st));
  HIP_TRY(hipStreamSynchronize(h->st));
  return 0;
}

int ganmf_stream_timer(ganmf_handle* h, int stop, double* ms) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  if (!h->ev_t0) { HIP_TRY(hipEventCreate(&h->ev_t0)); HIP_TRY(hipEventCreate(&h->ev_t1)); }
  if (!stop) { HIP_TRY(hipEventRecord(h->ev_t0, h->st)); return 0; }
  if (!ms) return fail(-1, "ganmf_stream_timer: null result pointer");
  HIP_TRY(hipEventRecord(h->ev_t1, h->st));
  HIP_TRY(hipEventSynchronize(h->ev_t1));
  float f = 0.f;
  HIP_TRY(hipEventElapsedTime(&f, h->ev_t0, h->ev_t1));
  *ms = (double)f;
  return 0;
}

int ganmf_profile_enable(ganmf_handle* h, int on) {
  if (!h) return fail(-1, "null handle");
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  for (auto& r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  h->recs.clear();
  h->prof = on != 0;
  return 0;
}

int ganmf_profile_read(ganmf_handle* h, ganmf_prof_entry* out, int32_t cap, int32_t* n_out) {
  if (!h || !out || !n_out) return fail(-1, "null argument");
  HIP_TRY(hipSetDevice(h->dev));
  HIP_TRY(hipStreamSynchronize(h->st));
  std::vector<ganmf_prof_entry> acc(T_COUNT);
  for (int i = 0; i < T_COUNT; ++i) {
    memset(&acc[i], 0, sizeof acc[i]);
    snprintf(acc[i].name, sizeof acc[i].name, "%s", kTagName[i]);
  }
  for (auto& r : h->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    acc[r.tag].launches += 1; acc[r.tag].ms += ms; acc[r.tag].flops += r.flops; acc[r.tag].bytes += r.bytes;
  }
  int n = 0;
  for (int i = 0; i < T_COUNT && n < cap; ++i)
    if (acc[i].launches) out[n++] = acc[i];
  *n_out = n;
  return 0;
}

int ganmf_gemm_f32(int device, const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K,
                   int a_kmajor, int b_kmajor, int tile, int nsplit, int iters, float* ms) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1) return fail(-1, "ganmf_gemm_f32: bad argument");
  if (a_kmajor && !b_kmajor) return fail(-1, "ganmf_gemm_f32: the TT layout is not part of the GANMF path");
  if (tile != 0 && tile != 64 && tile != 128) return fail(-1, "ganmf_gemm_f32: tile must be 0, 64 or 128");
  HIP_TRY(hipSetDevice(device));
  const int ar = a_kmajor ? K : M, ac = a_kmajor ? M : K;
  const int br = b_kmajor ? K : N, bc = b_kmajor ? N : K;
  const int lda = round_up(ac, LD_ALIGN), ldb = round_up(bc, LD_ALIGN), ldc = round_up((int)N, LD_ALIGN);
  float *dA = nullptr, *dB = nullptr, *dC = nullptr, *slab = nullptr, *zp = nullptr;
  TRY(dalloc(&zp, 2048 + 64));
  TRY(dalloc(&dA, (size_t)ar * lda)); TRY(dalloc(&dB, (size_t)br * ldb)); TRY(dalloc(&dC, (size_t)M * ldc));
  HIP_TRY(hipMemcpy2D(dA, (size_t)lda * 4, A, (size_t)ac * 4, (size_t)ac * 4, ar, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy2D(dB, (size_t)ldb * 4, B, (size_t)bc * 4, (size_t)bc * 4, br, hipMemcpyHostToDevice));
  GemmP g{};
  g.A = dA; g.lda = lda; g.B = dB; g.ldb = ldb; g.C = dC; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.nbatch = 1; g.epi.kind = EPI_STORE; g.zero_page = zp; g.c_pad_writable = 1;
  GemmTune tune;
  tune.tile = tile; tune.nsplit = nsplit;
  tune.mode = env_mfma_mode(MFMA_DEFAULT);
  tune.ring = env_int("GANMF_RING", 0);
  if (tune.ring != 0 && tune.ring != 2 && tune.ring != 3 && tune.ring != 4) tune.ring = 0;
  tune.persist = env_int("GANMF_PERSIST", -1);
  tune.kg = env_int("GANMF_KG", 0);
  tune.tile_order = env_int("GANMF_TILE_ORDER", 0);
  if (tune.kg != 0 && tune.kg != 1 && tune.kg != 2 && tune.kg != 4) tune.kg = 0;
  GemmPlan pl = gemm_plan(g.M, g.N, g.K, 1, false, tune);
  pl.persist = gemm_persist_eligible(g, a_kmajor, b_kmajor, pl, tune.persist) ? (tune.persist >= 2 ? tune.persist : 1) : 0;
  if (env_int("GANMF_X3KG", 0) && pl.mode == MFMA_F32 && pl.tile == 64 && pl.kg == 4 && pl.ring == 3 && !pl.persist) pl.mode = MFMA_BF16X3;      // (tests: the 16-wave split-bf16 kernel)
  const size_t slab_elems = gemm_slab_elems(pl, g.M, ldc, 1);
  if (slab_elems) TRY(dalloc(&slab, slab_elems));
  unsigned* counters = nullptr;
  const bool inl = env_int("GANMF_INKERNEL_REDUCE", 0) != 0;
  if (inl) TRY(dalloc((float**)&counters, COUNTER_CAP));
  hipStream_t st = nullptr;
  HIP_TRY(gemm_run(st, g, a_kmajor, b_kmajor, pl, slab, slab_elems, counters, COUNTER_CAP));
  HIP_TRY(hipDeviceSynchronize());
  if (iters > 1 || ms) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int i = 0; i < std::max(iters, 1); ++i) gemm_run(st, g, a_kmajor, b_kmajor, pl, slab, slab_elems, counters, COUNTER_CAP);
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float t = 0.f;
    hipEventElapsedTime(&t, a, b);
    if (ms) *ms = t / std::max(iters, 1);
    hipEventDestroy(a); hipEventDestroy(b);
  }
  HIP_TRY(hipMemcpy2D(C, (size_t)N * 4, dC, (size_t)ldc * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(zp);
  if (slab) hipFree(slab);
  if (counters) hipFree(counters);
  return 0;
}

}  // extern "C"
