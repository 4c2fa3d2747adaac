// fpc_engine.cpp -- host side of the C-ABI declared in include/fpc_engine.h.
//
// Owns every device allocation (tree SoA pools, board pools, staging), launches the kernels of
// fpc_tree_kernels.h / fpc_nn_kernels.h on one HIP stream, and never computes game logic on the
// host: positions are uploaded, processed by one wavefront each, and read back.  Compiled by hipcc
// for gfx950 into libfpc_engine.so (the product).  The same file builds against the wavefront
// emulator (-DFPC_EMUL, tests/emul/) for the CPU test-suite; that build has no NN and is never
// loaded by the product modules.
#include "fpc_tree_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#ifndef FPC_EMUL
#include <dlfcn.h>

#include "fpc_nn.h"
#endif

using namespace fpc;

namespace {
std::string g_create_error;

struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
};
}  // namespace

struct fpc_engine {
  fpc_config cfg{};
  DevCfg dc{};
  std::string err;
  hipStream_t stream = nullptr;
  // ---- batched board-op scratch
  int cap_boards = 0;
  fpc_board *d_boards = nullptr, *d_boards2 = nullptr;
  fpc_move *d_moves = nullptr;
  int *d_counts = nullptr, *d_results = nullptr, *d_err = nullptr, *d_player = nullptr, *d_flat = nullptr;
  float *d_dense = nullptr;       // encode / mask output staging [cap_boards * max(24*RR, A)]
  // ---- search
  Tree t{};
  std::vector<void *> allocs;
  int G = 0;
  double Cpuct = 0;
  bool searching = false;
  double *d_logtab = nullptr;
  float *d_enc_f32 = nullptr;     // [max_games,24,R,R]
  float *d_stats = nullptr;       // [max_games][SM_MAXCH][SM_REC] softmax chunk statistics of external logits
  fpc_board *d_roots = nullptr;   // staging [max_games]
  int *d_rc_i = nullptr;          // root-children gather: [2][rc_cap] flat | visits
  float *d_rc_f = nullptr;
  double *d_rc_d = nullptr;
  int *d_rc_meta = nullptr;       // [max_games][3]
  size_t rc_cap = 0;
  long sims_issued = 0;           // simulation steps since fpc_search_begin (bounded by max_sims: pools + log table)
  // ---- training tuples (device resident until the episode ends) and their RCCL exchange
  fpc_tuple *d_tuples = nullptr;
  int tuple_cap = 0, tuple_count = 0;
  int *d_tgame = nullptr;           // [max_games] game ids of one collect / set_z call
  float *d_tz = nullptr;            // [2][max_games]
  float *d_noise = nullptr;         // [max_games][FPC_MAX_MOVES] root-noise gamma draws (N4)
  int noise_n = 0;                  // rows uploaded by fpc_search_set_root_noise: a search must have exactly that many games
  void *comm = nullptr;             // ncclComm_t
  int comm_rank = 0, comm_world = 1;
  fpc_tuple *d_gather = nullptr;    // [world][gather_stride]
  long long *d_gcounts = nullptr;   // [3 * world + 4]: all-gathered (count, capacity, epoch) records | this rank's own record | its status word
  long long comm_epoch = 0;         // exchanges started on this communicator (tags every record and status word)
  int comm_fault = 0;               // fpc_debug_comm_fault
  int gcounts_world = 0;
  int gather_stride = 0;
  size_t gather_cap = 0;            // tuples the receive buffer holds (world x padded count)
  std::vector<int> gather_counts;
  // ---- stats
  bool timing = false;
  int policy_mode = 0;            // FPC_POLICY_FULL / FPC_POLICY_LEGAL (fpc_search_run only)
  fpc_stats stats{};
  // asynchronous stage timing: events are only RECORDED on the stream during the search and read
  // back in fpc_search_results, so enabling it does not serialise the pipeline.  Five events per
  // simulation step still cost 2.7 % of the step, so only every TIMING_PERIOD-th step carries them
  // and its intervals are scaled by the period.
  static constexpr int TIMING_PERIOD = 16;   // round 3: at 8 the events still cost 0.3-1.3 % of the step (same-box A/B against --no-stage-timing)
  uint64_t tstep = 0;
  bool tsample = false;
  std::vector<hipEvent_t> evpool;
  std::vector<int> evtag;          // 0 step start, 1 tower start, 2 Linear start, 3 expand start, 4 step end
  size_t evused = 0;
#ifndef FPC_EMUL
  fpc::NN nn;
#endif
};

namespace {

int fail(fpc_engine *e, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
}

// every entry point re-selects the handle's device: one process may hold engines on several GPUs
#define USE_DEV(e)                                                                             \
  do {                                                                                         \
    if ((e) && hipSetDevice((e)->cfg.device) != hipSuccess) return fail(e, FPC_ENODEVICE, "hipSetDevice(%d) failed", (e)->cfg.device); \
  } while (0)

#define HIPCHK(e, call)                                                                        \
  do {                                                                                         \
    hipError_t _r = (call);                                                                    \
    if (_r != hipSuccess) return fail(e, FPC_ENODEVICE, "%s failed: %s", #call, hipGetErrorString(_r)); \
  } while (0)

template <class T>
int dalloc(fpc_engine *e, T **p, size_t count) {
  void *q = nullptr;
  if (hipMalloc(&q, count * sizeof(T)) != hipSuccess || !q) return fail(e, FPC_ENOMEM, "hipMalloc of %zu bytes failed", count * sizeof(T));
  if (hipMemset(q, 0, count * sizeof(T)) != hipSuccess) return fail(e, FPC_ENODEVICE, "hipMemset failed");
  e->allocs.push_back(q);
  *p = (T *)q;
  return 0;
}

int ensure_board_scratch(fpc_engine *e, int n) {
  if (n <= e->cap_boards) return 0;
  int cap = std::max(n, std::max(64, e->cap_boards * 2));
  auto re = [&](auto **p, size_t count) -> int {
    if (*p) { (void)hipFree(*p); e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), (void *)*p)); *p = nullptr; }
    return dalloc(e, p, count);
  };
  int r;
  if ((r = re(&e->d_boards, cap))) return r;
  if ((r = re(&e->d_boards2, cap))) return r;
  if ((r = re(&e->d_moves, (size_t)cap * FPC_MAX_MOVES))) return r;
  if ((r = re(&e->d_counts, cap))) return r;
  if ((r = re(&e->d_results, cap))) return r;
  if ((r = re(&e->d_err, cap))) return r;
  if ((r = re(&e->d_player, cap))) return r;
  if ((r = re(&e->d_flat, cap))) return r;
  if ((r = re(&e->d_dense, (size_t)cap * std::max(24 * e->dc.RR, e->dc.A)))) return r;
  e->cap_boards = cap;
  return 0;
}

int check_boards(fpc_engine *e, const fpc_board *b, int n) {
  for (int i = 0; i < n; ++i) {
    for (int c = 0; c < 4; ++c) {
      if (b[i].castle[c] > 3) return fail(e, FPC_EINVAL, "board %d: bad castling rights", i);
      if (b[i].plen[c] > FPC_MAX_PL) return fail(e, FPC_EINVAL, "board %d: piece list longer than %d", i, FPC_MAX_PL);
    }
    if (b[i].turn > 3) return fail(e, FPC_EINVAL, "board %d: bad turn", i);
  }
  return 0;
}

int err_to_status(fpc_engine *e, const std::vector<int> &errs, const char *what) {
  for (size_t i = 0; i < errs.size(); ++i) {
    const int x = errs[i];
    if (!x) continue;
    if (x & ERR_MOVE) return fail(e, FPC_EMOVE, "%s %zu: piece missing for move (engine/board.cpp:1046-1054)", what, i);
    if (x & ERR_SELECT) return fail(e, FPC_ESELECT, "%s %zu: Failed to select a child. (node.cpp:72-75)", what, i);
    if (x & ERR_POLICY) return fail(e, FPC_EPOLICY, "%s %zu: legal policy mass is zero/NaN (reference expands every index and throws)", what, i);
    return fail(e, FPC_ECAPACITY, "%s %zu: capacity exceeded (bits %d: moves>%d / node pool / board pool)", what, i, x, FPC_MAX_MOVES);
  }
  return 0;
}

int run_board_ops(fpc_engine *e, fpc_board *boards, int n, int ops, const int *player, const int *flat) {
  int r;
  if ((r = ensure_board_scratch(e, n))) return r;
  HIPCHK(e, hipMemcpyAsync(e->d_boards, boards, (size_t)n * sizeof(fpc_board), hipMemcpyHostToDevice, e->stream));
  if (player) HIPCHK(e, hipMemcpyAsync(e->d_player, player, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  if (flat) HIPCHK(e, hipMemcpyAsync(e->d_flat, flat, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  FPC_LAUNCH(k_board_ops, n, 64, e->stream, e->dc, e->d_boards, n, ops, player ? e->d_player : (const int *)nullptr,
             flat ? e->d_flat : (const int *)nullptr, e->d_moves, e->d_counts, e->d_results, e->d_err);
  HIPCHK(e, hipGetLastError());
  return 0;
}

int fetch_errs(fpc_engine *e, int n, const char *what) {
  std::vector<int> errs(n);
  HIPCHK(e, hipMemcpyAsync(errs.data(), e->d_err, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return err_to_status(e, errs, what);
}

struct LocHash {   // std::hash<chess::BoardLocation>, engine/board.h:229-237
  int R;
  size_t operator()(uint8_t s) const {
    size_t h = 14479 + 14593 * (size_t)(int8_t)(s / R);
    h += 24439 * (size_t)(int8_t)(s % R);
    return h;
  }
};

void mark(fpc_engine *e, int tag) {
  if (tag == 0) e->tsample = e->timing && (e->tstep++ % fpc_engine::TIMING_PERIOD) == 0;
  if (!e->tsample) return;
  if (e->evused == e->evpool.size()) {
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return;
    e->evpool.push_back(ev);
    e->evtag.push_back(0);
  }
  e->evtag[e->evused] = tag;
  (void)hipEventRecord(e->evpool[e->evused], e->stream);
  e->evused++;
}
// call after the stream has been synchronised
void resolve_marks(fpc_engine *e) {
  for (size_t i = 0; i + 1 < e->evused; ++i) {
    const int a = e->evtag[i], b = e->evtag[i + 1];
    if (b != a + 1) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->evpool[i], e->evpool[i + 1]) != hipSuccess) continue;
    ms *= (float)fpc_engine::TIMING_PERIOD;
    if (a == 0) e->stats.ms_select += ms; else if (a == 1) e->stats.ms_tower += ms; else if (a == 2) e->stats.ms_fc += ms; else e->stats.ms_expand += ms;
  }
  e->evused = 0;
}

}  // namespace

extern "C" {

int fpc_abi_version(void) { return 7; }

const char *fpc_last_error(const fpc_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int fpc_num_action_channels(int R) { return 8 * R + 8; }
int fpc_action_space_size(int R) { return (8 * R + 8) * R * R; }
int fpc_is_legal_location(int R, int INV, int row, int col) {
  const int mx = R - 1;
  if (row < 0 || row > mx || col < 0 || col > mx) return 0;
  const bool cc = col < INV || col > mx - INV;
  return !(cc && (row < INV || row > mx - INV));
}
int fpc_move_flat_index(int R, int from, int to) {
  if (from < 0 || to < 0 || from >= R * R || to >= R * R || from == to) return -1;
  const int dy = to / R - from / R, dx = to % R - from % R;
  const int ay = std::abs(dy), ax = std::abs(dx);
  const bool queen = dx == 0 || dy == 0 || ax == ay;
  const bool knight = (ax == 1 && ay == 2) || (ax == 2 && ay == 1);
  if (!queen && !knight) return -1;
  DevCfg c = make_devcfg(R, 0, 0);
  return move_plane(c, from, to) * R * R + from;
}
int fpc_flat_to_move(int R, int flat, int *from, int *to) {
  DevCfg c = make_devcfg(R, 0, 0);
  if (flat < 0 || flat >= c.A) return FPC_EINVAL;
  *to = flat_to(c, flat, from);
  return 0;
}

int fpc_board_from_dict(fpc_board *out, int R, int turn, const uint8_t *sq, const uint8_t *piece, int n,
                        const uint8_t *castle4) {
  if (!out || R < 4 || R * R > FPC_MAX_SQ || turn < 0 || turn > 3 || n < 0) return FPC_EINVAL;
  memset(out, 0, sizeof(*out));
  for (int c = 0; c < 4; ++c) out->king[c] = FPC_NO_SQ;
  out->turn = (uint8_t)turn;
  if (castle4) for (int c = 0; c < 4; ++c) out->castle[c] = castle4[c] & 3;
  // Same container, hash and insertion sequence as the reference constructor sees (pybind11's dict
  // -> unordered_map caster reserves, then emplaces in dict order), so iteration order -- which
  // the reference bakes into piece_list_ (engine/board.cpp:1209-1223) -- is reproduced.
  std::unordered_map<uint8_t, uint8_t, LocHash> m(0, LocHash{R});
  m.reserve((size_t)n);
  for (int i = 0; i < n; ++i) {
    if (sq[i] >= R * R || !(piece[i] & 0x80)) return FPC_EINVAL;
    m.emplace(sq[i], piece[i]);
  }
  std::vector<std::pair<uint8_t, uint8_t>> lists[4];
  for (const auto &it : m) {
    out->sq[it.first] = it.second;
    const int col = (it.second >> 5) & 3;
    lists[col].push_back({it.first, it.second});
    if (((it.second >> 2) & 7) == KING) out->king[col] = it.first;
  }
  static const int score[8] = {1, 2, 3, 4, 5, 0, 0, 0};   // piece_move_order_scores, engine/board.cpp:1230-1236
  for (int c = 0; c < 4; ++c) {
    std::sort(lists[c].begin(), lists[c].end(), [](const std::pair<uint8_t, uint8_t> &a, const std::pair<uint8_t, uint8_t> &b) {
      return score[(a.second >> 2) & 7] < score[(b.second >> 2) & 7];
    });
    if (lists[c].size() > FPC_MAX_PL) return FPC_ECAPACITY;
    out->plen[c] = (uint8_t)lists[c].size();
    for (size_t i = 0; i < lists[c].size(); ++i) out->pl[c][i] = lists[c][i].first;
  }
  return 0;
}

int fpc_board_heuristic(const fpc_board *b, int team) {   // engine/board.cpp:1263-1292
  static const int val[8] = {1, 3, 3, 5, 9, 0, 0, 0};
  int h = 0;
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < b->plen[c] && i < FPC_MAX_PL; ++i) {
      const uint8_t p = b->sq[b->pl[c][i]];
      if (!(p & 0x80)) continue;
      const int ty = (p >> 2) & 7;
      if (ty == KING) continue;
      h += (((p >> 5) & 1) == team) ? val[ty] : -val[ty];
    }
  return h;
}

int fpc_create(const fpc_config *cfg, fpc_engine **out) {
  if (!cfg || !out) return fail(nullptr, FPC_EINVAL, "null argument");
  const int R = cfg->board_size, INV = cfg->invalid_area;
  // geometries the kernels implement: 8 generator slots per piece bound the knight loop (2*2*(INV-1) <= 8,
  // engine/board.cpp:188) and the network kernels tile boards of 8..14 squares a side
  if (R < 8 || R > 14 || INV < 1 || INV > 3 || 2 * INV >= R) return fail(nullptr, FPC_EINVAL, "unsupported board %dx%d/%d (supported: 8..14 a side, cut corners 1..3)", R, R, INV);
  if (cfg->max_games < 1 || cfg->max_sims < 1) return fail(nullptr, FPC_EINVAL, "max_games/max_sims must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device || cfg->device < 0)
    return fail(nullptr, FPC_ENODEVICE, "no HIP device %d (found %d): this engine has no CPU path", cfg->device, ndev);
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, FPC_ENODEVICE, "hipSetDevice(%d) failed", cfg->device);
  fpc_engine *e = new fpc_engine();
  e->cfg = *cfg;
  if (e->cfg.avg_children <= 0) e->cfg.avg_children = 96;
  e->dc = make_devcfg(R, INV, FPC_RULES_STRICT);
  int r = 0;
  auto bail = [&](int code) { g_create_error = e->err; fpc_destroy(e); return code; };
  if (hipStreamCreate(&e->stream) != hipSuccess) { e->err = "hipStreamCreate failed"; return bail(FPC_ENODEVICE); }
  const int Gm = cfg->max_games;
  Tree &t = e->t;
  t.node_cap = 1 + cfg->max_sims * e->cfg.avg_children;
  t.board_cap = cfg->max_sims + 2;
  t.path_cap = cfg->max_sims + 2;          // a descent visits at most one node per earlier expansion + the root
  const size_t nn = (size_t)Gm * t.node_cap;
  if ((r = dalloc(e, &t.N, nn)) || (r = dalloc(e, &t.W, nn)) || (r = dalloc(e, &t.P, nn)) || (r = dalloc(e, &t.mv, nn)) ||
      (r = dalloc(e, &t.parent, nn)) || (r = dalloc(e, &t.child0, nn)) || (r = dalloc(e, &t.nch, nn)) ||
      (r = dalloc(e, &t.bslot, nn)) || (r = dalloc(e, &t.boards, (size_t)Gm * t.board_cap)) ||
      (r = dalloc(e, &t.nnodes, Gm)) || (r = dalloc(e, &t.nboards, Gm)) || (r = dalloc(e, &t.alive, Gm)) ||
      (r = dalloc(e, &t.sims_done, Gm)) || (r = dalloc(e, &t.err, Gm)) || (r = dalloc(e, &t.leaf_node, Gm)) || (r = dalloc(e, &t.leaf_node_nx, Gm)) ||
      (r = dalloc(e, &t.leaf_slot_nx, Gm)) || (r = dalloc(e, &t.leaf_turn_nx, Gm)) ||
      (r = dalloc(e, &t.leaf_turn, Gm)) || (r = dalloc(e, &t.nlegal, Gm)) ||
      (r = dalloc(e, &t.path, (size_t)Gm * t.path_cap)) || (r = dalloc(e, &t.path_len, Gm)) ||
      (r = dalloc(e, &t.legal, (size_t)Gm * FPC_MAX_MOVES)) || (r = dalloc(e, &t.leaf_slot, Gm)) ||
      (r = dalloc(e, &e->d_enc_f32, (size_t)Gm * 24 * e->dc.RR)) || (r = dalloc(e, &e->d_stats, (size_t)Gm * SM_MAXCH * SM_REC)) || (r = dalloc(e, &e->d_roots, Gm)) ||
      (r = dalloc(e, &e->d_logtab, (size_t)cfg->max_sims + 16)) ||
      (r = dalloc(e, &e->d_rc_meta, (size_t)Gm * 3)))
    return bail(r);
  {
    // log(sqrt(N_parent)) (node.cpp:53-54) tabulated with the HOST libm so that the device PUCT
    // sees exactly the bits the reference's std::log produces; +,*,/,sqrt are IEEE on both sides.
    std::vector<double> lt((size_t)cfg->max_sims + 16);
    for (size_t i = 0; i < lt.size(); ++i) lt[i] = i == 0 ? 0.0 : std::log(std::sqrt((double)i));
    if (hipMemcpy(e->d_logtab, lt.data(), lt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      e->err = "logtab upload failed";
      return bail(FPC_ENODEVICE);
    }
  }
#ifndef FPC_EMUL
  if ((r = e->nn.init(e->dc, Gm, cfg->nn_dtype, e->stream, &e->err))) return bail(r);
#endif
  *out = e;
  return 0;
}

void fpc_destroy(fpc_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  (void)fpc_comm_destroy(e);
#ifndef FPC_EMUL
  e->nn.destroy();
#endif
  for (void *p : e->allocs) (void)hipFree(p);
  for (auto &ev : e->evpool) if (ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// ------------------------------------------------------------------------------------------------
int fpc_boards_legal_moves(fpc_engine *e, fpc_board *boards, int n, fpc_move *moves, int *counts) {
  if (!e || !boards || !moves || !counts || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n == 0) return 0;
  int r;
  if ((r = check_boards(e, boards, n))) return r;
  if ((r = run_board_ops(e, boards, n, OP_LEGAL, nullptr, nullptr))) return r;
  HIPCHK(e, hipMemcpyAsync(boards, e->d_boards, (size_t)n * sizeof(fpc_board), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(counts, e->d_counts, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(moves, e->d_moves, (size_t)n * FPC_MAX_MOVES * sizeof(fpc_move), hipMemcpyDeviceToHost, e->stream));
  return fetch_errs(e, n, "board");
}

int fpc_boards_game_result(fpc_engine *e, fpc_board *boards, int n, const int *player, int *results) {
  if (!e || !boards || !results || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n == 0) return 0;
  int r;
  if ((r = check_boards(e, boards, n))) return r;
  if ((r = run_board_ops(e, boards, n, OP_RESULT, player, nullptr))) return r;
  HIPCHK(e, hipMemcpyAsync(boards, e->d_boards, (size_t)n * sizeof(fpc_board), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(results, e->d_results, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  return fetch_errs(e, n, "board");
}

int fpc_boards_take_action(fpc_engine *e, const fpc_board *boards, const int *flat, int n, fpc_board *out) {
  if (!e || !boards || !flat || !out || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n == 0) return 0;
  int r;
  if ((r = check_boards(e, boards, n))) return r;
  for (int i = 0; i < n; ++i)
    if (flat[i] < 0 || flat[i] >= e->dc.A) return fail(e, FPC_EINVAL, "flat index %d out of range", flat[i]);
  if ((r = run_board_ops(e, const_cast<fpc_board *>(boards), n, OP_TAKE, nullptr, flat))) return r;
  HIPCHK(e, hipMemcpyAsync(out, e->d_boards, (size_t)n * sizeof(fpc_board), hipMemcpyDeviceToHost, e->stream));
  return fetch_errs(e, n, "board");
}

int fpc_boards_attack_maps(fpc_engine *e, const fpc_board *boards, int n, uint8_t *out_host) {
  if (!e || !boards || !out_host || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n == 0) return 0;
  int r;
  if ((r = check_boards(e, boards, n))) return r;
  if ((r = ensure_board_scratch(e, n))) return r;
  HIPCHK(e, hipMemcpyAsync(e->d_boards, boards, (size_t)n * sizeof(fpc_board), hipMemcpyHostToDevice, e->stream));
  uint8_t *d_maps = reinterpret_cast<uint8_t *>(e->d_dense);       // [n][24][RR] f32 of scratch: room for [n][6][RR] bytes
  FPC_LAUNCH(k_attack_maps, n, 64, e->stream, e->dc, (const fpc_board *)e->d_boards, n, d_maps);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(out_host, d_maps, (size_t)n * 6 * e->dc.RR, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return 0;
}

int fpc_boards_encode(fpc_engine *e, const fpc_board *boards, int n, float *out_host) {
  if (!e || !boards || !out_host || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n == 0) return 0;
  int r;
  if ((r = check_boards(e, boards, n))) return r;
  if ((r = ensure_board_scratch(e, n))) return r;
  HIPCHK(e, hipMemcpyAsync(e->d_boards, boards, (size_t)n * sizeof(fpc_board), hipMemcpyHostToDevice, e->stream));
  FPC_LAUNCH(k_encode, n, 64, e->stream, e->dc, (const fpc_board *)e->d_boards, 1, (const int *)nullptr,
             (const int *)nullptr, n, 0, e->d_dense, (uint16_t *)nullptr, (uint16_t)0, (int)boards[0].turn);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(out_host, e->d_dense, (size_t)n * 24 * e->dc.RR * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return 0;
}

int fpc_boards_legal_mask(fpc_engine *e, fpc_board *boards, int n, float *out_host) {
  if (!e || !boards || !out_host || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n == 0) return 0;
  int r;
  if ((r = check_boards(e, boards, n))) return r;
  if ((r = run_board_ops(e, boards, n, OP_LEGAL, nullptr, nullptr))) return r;
  FPC_LAUNCH(k_mask_from_moves, n, 64, e->stream, e->dc, (const fpc_move *)e->d_moves, (const int *)e->d_counts, n, e->d_dense);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(boards, e->d_boards, (size_t)n * sizeof(fpc_board), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(out_host, e->d_dense, (size_t)n * e->dc.A * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  return fetch_errs(e, n, "board");
}

// ------------------------------------------------------------------------------------------------
int fpc_search_begin(fpc_engine *e, const fpc_board *roots, int n_games, double c_puct) {
  if (!e || !roots || n_games < 1) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  if (n_games > e->cfg.max_games) return fail(e, FPC_EINVAL, "n_games %d > max_games %d", n_games, e->cfg.max_games);
  if (e->t.noise && e->noise_n != n_games)
    return fail(e, FPC_EINVAL, "root noise was uploaded for %d games, this search has %d: call fpc_search_set_root_noise again (or with NULL)", e->noise_n, n_games);
  int r;
  if ((r = check_boards(e, roots, n_games))) return r;
  e->G = n_games;
  e->Cpuct = c_puct;
  HIPCHK(e, hipMemcpyAsync(e->d_roots, roots, (size_t)n_games * sizeof(fpc_board), hipMemcpyHostToDevice, e->stream));
  FPC_LAUNCH(k_search_init, n_games, 64, e->stream, e->t, n_games, (const fpc_board *)e->d_roots);
  HIPCHK(e, hipGetLastError());
  e->searching = true;
  e->sims_issued = 0;
  return 0;
}

static int launch_select(fpc_engine *e) {
  mark(e, 0);
  FPC_LAUNCH(k_select, e->G, 64, e->stream, e->dc, e->t, e->G, e->Cpuct, (const double *)e->d_logtab);
  return 0;
}

// softmax chunk statistics of logits that did not come from the internal policy Linear
static void launch_partials(fpc_engine *e, const float *logits_dev) {
  const int nchunks = (e->dc.A / 4 + SM_THREADS - 1) / SM_THREADS;
  FPC_LAUNCH(k_softmax_partials, e->G * nchunks, SM_THREADS, e->stream, logits_dev, e->dc.A, e->G, nchunks, e->d_stats);
}

// after a k_expand_select launch the leaves it selected become the current ones
static void swap_leaf_arrays(fpc_engine *e) {
  std::swap(e->t.leaf_node, e->t.leaf_node_nx);
  std::swap(e->t.leaf_slot, e->t.leaf_slot_nx);
  std::swap(e->t.leaf_turn, e->t.leaf_turn_nx);
}

static int finish_select(fpc_engine *e, int *n_live, const float **enc_dev) {
  FPC_LAUNCH(k_encode, e->G, 64, e->stream, e->dc, (const fpc_board *)e->t.boards, e->t.board_cap,
             (const int *)e->t.leaf_slot, (const int *)e->t.leaf_turn, e->G, 0, e->d_enc_f32, (uint16_t *)nullptr,
             (uint16_t)0, -1);
  HIPCHK(e, hipGetLastError());
  mark(e, 1);
  e->stats.launches_select++;
  std::vector<int> slots(e->G);
  HIPCHK(e, hipMemcpyAsync(slots.data(), e->t.leaf_slot, (size_t)e->G * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  int live = 0;
  for (int s : slots) live += s >= 0;
  if (n_live) *n_live = live;
  if (enc_dev) *enc_dev = e->d_enc_f32;
  return 0;
}

int fpc_search_select(fpc_engine *e, int *n_live, const float **enc_dev) {
  if (!e || !e->searching) return fail(e, FPC_ESTATE, "fpc_search_begin has not been called");
  USE_DEV(e);
  if (e->sims_issued + 1 > e->cfg.max_sims) return fail(e, FPC_ECAPACITY, "more than max_sims = %d simulations since fpc_search_begin", e->cfg.max_sims);
  e->sims_issued += 1;
  launch_select(e);
  return finish_select(e, n_live, enc_dev);
}

int fpc_search_expand_select(fpc_engine *e, const float *logits_dev, const float *value_dev, int *n_live, const float **enc_dev) {
  if (!e || !e->searching) return fail(e, FPC_ESTATE, "fpc_search_begin has not been called");
  USE_DEV(e);
  if (!logits_dev || !value_dev) return fail(e, FPC_EINVAL, "null logits/value");
  if (e->sims_issued + 1 > e->cfg.max_sims) return fail(e, FPC_ECAPACITY, "more than max_sims = %d simulations since fpc_search_begin", e->cfg.max_sims);
  e->sims_issued += 1;
  mark(e, 3);
  launch_partials(e, logits_dev);
  FPC_LAUNCH(k_expand_select, e->G, EXPAND_THREADS, e->stream, e->dc, e->t, e->G, (LogitSrc{logits_dev, nullptr, nullptr, 0, 0, 0, 0, 256}),
             (const float *)e->d_stats, value_dev, e->Cpuct, (const double *)e->d_logtab);
  HIPCHK(e, hipGetLastError());
  swap_leaf_arrays(e);
  mark(e, 4);
  e->stats.launches_expand++;
  mark(e, 0);
  return finish_select(e, n_live, enc_dev);
}

int fpc_search_expand(fpc_engine *e, const float *logits_dev, const float *value_dev) {
  if (!e || !e->searching) return fail(e, FPC_ESTATE, "fpc_search_begin has not been called");
  USE_DEV(e);
  if (!logits_dev || !value_dev) return fail(e, FPC_EINVAL, "null logits/value");
  mark(e, 3);
  launch_partials(e, logits_dev);
  FPC_LAUNCH(k_expand, e->G, EXPAND_THREADS, e->stream, e->dc, e->t, e->G, (LogitSrc{logits_dev, nullptr, nullptr, 0, 0, 0, 0, 256}), (const float *)e->d_stats, value_dev);
  HIPCHK(e, hipGetLastError());
  mark(e, 4);
  e->stats.launches_expand++;
  return 0;
}

int fpc_search_run(fpc_engine *e, int sims) {
  if (!e || !e->searching) return fail(e, FPC_ESTATE, "fpc_search_begin has not been called");
  USE_DEV(e);
  if (sims < 0 || sims > e->cfg.max_sims) return fail(e, FPC_EINVAL, "sims %d > max_sims %d", sims, e->cfg.max_sims);
  if (e->sims_issued + sims > e->cfg.max_sims) return fail(e, FPC_ECAPACITY, "more than max_sims = %d simulations since fpc_search_begin", e->cfg.max_sims);
  e->sims_issued += sims;
#ifdef FPC_EMUL
  return fail(e, FPC_EWEIGHTS, "the internal ResNet exists only in the gfx950 build");
#else
  if (!e->nn.loaded) return fail(e, FPC_EWEIGHTS, "fpc_load_weights has not been called");
  // step s: [k_select] -> network -> k_expand; for all but the last step of this call the expansion and the
  // NEXT step's selection are one launch (k_expand_select), so a step is 4 dependent kernels instead of 5
  for (int s = 0; s < sims; ++s) {
    if (s == 0) launch_select(e); else mark(e, 0);
    if (e->nn.takes_boards())      // the tower megakernel encodes its games' leaves itself
      e->nn.set_board_input((const fpc_board *)e->t.boards, e->t.board_cap, (const int *)e->t.leaf_slot, (const int *)e->t.leaf_turn);
    else
      FPC_LAUNCH(k_encode, e->G, 64, e->stream, e->dc, (const fpc_board *)e->t.boards, e->t.board_cap,
                 (const int *)e->t.leaf_slot, (const int *)e->t.leaf_turn, e->G, 1, (float *)nullptr, e->nn.input16(),
                 e->nn.one16(), -1);
    mark(e, 1);
    e->nn.mark_fn = [](void *ctx, int tag) { mark((fpc_engine *)ctx, tag); };
    e->nn.mark_ctx = e;
    const bool dense = getenv("FPC_DEV_KNOBS") && atoi(getenv("FPC_DEV_KNOBS")) != 0 && getenv("FPC_DENSE_LOGITS") != nullptr;    // developer knob (A/B, with FPC_DEV_KNOBS=1): the dense [G][A] logits matrix is written as well
    int r = e->policy_mode == FPC_POLICY_LEGAL ? e->nn.forward_legal(e->G, e->t, &e->err) : e->nn.forward(e->G, dense, &e->err);
    e->nn.mark_fn = nullptr;
    if (r) return r;
    mark(e, 3);
    const bool fuse = s + 1 < sims;
    if (e->policy_mode == FPC_POLICY_LEGAL) {
      if (fuse)
        FPC_LAUNCH(k_expand_legal_select, e->G, 64, e->stream, e->dc, e->t, e->G, (const float *)e->nn.legal_logits(), (const float *)e->nn.value(),
                   e->Cpuct, (const double *)e->d_logtab);
      else
        FPC_LAUNCH(k_expand_legal, e->G, 64, e->stream, e->dc, e->t, e->G, (const float *)e->nn.legal_logits(), (const float *)e->nn.value());
    } else {
      if (fuse)
        FPC_LAUNCH(k_expand_select, e->G, EXPAND_THREADS, e->stream, e->dc, e->t, e->G, e->nn.logit_src(dense), (const float *)e->nn.stats(),
                   (const float *)e->nn.value(), e->Cpuct, (const double *)e->d_logtab);
      else
        FPC_LAUNCH(k_expand, e->G, EXPAND_THREADS, e->stream, e->dc, e->t, e->G, e->nn.logit_src(dense), (const float *)e->nn.stats(),
                   (const float *)e->nn.value());
    }
    if (fuse) swap_leaf_arrays(e);
    mark(e, 4);
    e->stats.launches_select++; e->stats.launches_nn++; e->stats.launches_expand++;
  }
  HIPCHK(e, hipGetLastError());
  return 0;
#endif
}

int fpc_search_results(fpc_engine *e, fpc_board *roots_out, int *root_visits, int *n_children, int *sims_done,
                       int max_children, int *child_flat, int *child_visits, float *child_prior,
                       double *child_value_sum) {
  if (!e || !e->searching) return fail(e, FPC_ESTATE, "fpc_search_begin has not been called");
  USE_DEV(e);
  if (max_children < 0) return fail(e, FPC_EINVAL, "bad max_children");
  const int G = e->G;
  int r;
  const size_t need = (size_t)G * std::max(max_children, 1);
  if (need > e->rc_cap) {
    auto drop = [&](auto **p) {
      if (!*p) return;
      (void)hipFree(*p);
      e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), (void *)*p));
      *p = nullptr;
    };
    drop(&e->d_rc_i); drop(&e->d_rc_f); drop(&e->d_rc_d);
    e->rc_cap = 0;
    if ((r = dalloc(e, &e->d_rc_i, need * 2)) || (r = dalloc(e, &e->d_rc_f, need)) || (r = dalloc(e, &e->d_rc_d, need))) return r;
    e->rc_cap = need;
  }
  int *d_flat = e->d_rc_i, *d_vis = e->d_rc_i + need, *d_meta = e->d_rc_meta;
  FPC_LAUNCH(k_root_children, G, 64, e->stream, e->t, G, max_children, d_flat, d_vis, e->d_rc_f, e->d_rc_d, d_meta, e->d_roots);
  HIPCHK(e, hipGetLastError());
  std::vector<int> meta((size_t)G * 3), errs(G);
  HIPCHK(e, hipMemcpyAsync(meta.data(), d_meta, meta.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(errs.data(), e->t.err, (size_t)G * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  if (roots_out) HIPCHK(e, hipMemcpyAsync(roots_out, e->d_roots, (size_t)G * sizeof(fpc_board), hipMemcpyDeviceToHost, e->stream));
  if (child_flat) HIPCHK(e, hipMemcpyAsync(child_flat, d_flat, need * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  if (child_visits) HIPCHK(e, hipMemcpyAsync(child_visits, d_vis, need * sizeof(int), hipMemcpyDeviceToHost, e->stream));
  if (child_prior) HIPCHK(e, hipMemcpyAsync(child_prior, e->d_rc_f, need * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (child_value_sum) HIPCHK(e, hipMemcpyAsync(child_value_sum, e->d_rc_d, need * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  resolve_marks(e);
  uint64_t sims = 0, nodes = 0;
  for (int g = 0; g < G; ++g) {
    if (root_visits) root_visits[g] = meta[(size_t)g * 3 + 0];
    if (n_children) n_children[g] = meta[(size_t)g * 3 + 1];
    if (sims_done) sims_done[g] = meta[(size_t)g * 3 + 2];
    sims += (uint64_t)meta[(size_t)g * 3 + 2];
  }
  (void)nodes;
  e->stats.sims += sims;
  return err_to_status(e, errs, "game");
}

int fpc_search_grandchildren(fpc_engine *e, int game, int child_idx, int max_children, int *n, int *flat, int *visits) {
  if (!e || !e->searching || game < 0 || game >= e->G || !n) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  const size_t nb = (size_t)game * e->t.node_cap;
  int c0 = -1;
  uint16_t nc = 0;
  HIPCHK(e, hipMemcpy(&c0, e->t.child0 + nb, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(e, hipMemcpy(&nc, e->t.nch + nb, sizeof(uint16_t), hipMemcpyDeviceToHost));
  if (c0 < 0 || child_idx < 0 || child_idx >= nc) return fail(e, FPC_EINVAL, "no such root child");
  const size_t ch = nb + c0 + child_idx;
  int g0 = -1;
  uint16_t gn = 0;
  HIPCHK(e, hipMemcpy(&g0, e->t.child0 + ch, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(e, hipMemcpy(&gn, e->t.nch + ch, sizeof(uint16_t), hipMemcpyDeviceToHost));
  if (g0 < 0) { *n = 0; return 0; }
  *n = gn;
  const int k = std::min<int>(gn, max_children);
  std::vector<uint16_t> mv(k);
  HIPCHK(e, hipMemcpy(mv.data(), e->t.mv + nb + g0, (size_t)k * sizeof(uint16_t), hipMemcpyDeviceToHost));
  if (visits) HIPCHK(e, hipMemcpy(visits, e->t.N + nb + g0, (size_t)k * sizeof(int), hipMemcpyDeviceToHost));
  if (flat) for (int i = 0; i < k; ++i) flat[i] = mv[i];
  return 0;
}

// ------------------------------------------------------------------------------------------------
int fpc_load_weights(fpc_engine *e, const void *blob, uint64_t nbytes) {
  if (!e || !blob) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
#ifdef FPC_EMUL
  (void)nbytes;
  return fail(e, FPC_EWEIGHTS, "the internal ResNet exists only in the gfx950 build");
#else
  return e->nn.load(blob, nbytes, &e->err);
#endif
}

int fpc_nn_forward(fpc_engine *e, const float *enc_dev, int n, float *logits_dev, float *value_dev) {
  if (!e || !enc_dev || !logits_dev || !value_dev) return fail(e, FPC_EINVAL, "bad argument");
  USE_DEV(e);
#ifdef FPC_EMUL
  (void)n;
  return fail(e, FPC_EWEIGHTS, "the internal ResNet exists only in the gfx950 build");
#else
  if (!e->nn.loaded) return fail(e, FPC_EWEIGHTS, "fpc_load_weights has not been called");
  if (n < 1 || n > e->cfg.max_games) return fail(e, FPC_EINVAL, "n out of range");
  return e->nn.forward_external(enc_dev, n, logits_dev, value_dev, &e->err);
#endif
}

// ------------------------------------------------------------------------------------------------
// training tuples
int fpc_tuples_reserve(fpc_engine *e, int capacity) {
  if (!e || capacity < 0) return fail(e, FPC_EINVAL, "bad argument");
  // fpc_tuple carries a child's visit count (1 + its simulations, Q1) as u16
  if (e->cfg.max_sims > 65534) return fail(e, FPC_EUNSUPPORTED, "training tuples hold visit counts as u16: max_sims %d > 65534", e->cfg.max_sims);
  USE_DEV(e);
  if (e->d_tuples) {
    (void)hipFree(e->d_tuples);
    e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), (void *)e->d_tuples));
    e->d_tuples = nullptr;
  }
  e->tuple_cap = e->tuple_count = 0;
  int r;
  if (capacity > 0 && (r = dalloc(e, &e->d_tuples, (size_t)capacity))) return r;
  if (!e->d_tgame && ((r = dalloc(e, &e->d_tgame, (size_t)e->cfg.max_games)) || (r = dalloc(e, &e->d_tz, (size_t)2 * e->cfg.max_games)))) return r;
  e->tuple_cap = capacity;
  return 0;
}

int fpc_tuples_reset(fpc_engine *e) {
  if (!e) return FPC_EINVAL;
  e->tuple_count = 0;
  return 0;
}

int fpc_collect_tuples(fpc_engine *e, const int *game_id, int ply) {
  if (!e || !e->searching) return fail(e, FPC_ESTATE, "fpc_search_begin has not been called");
  USE_DEV(e);
  const int G = e->G;
  if (e->tuple_count + G > e->tuple_cap) return fail(e, FPC_ECAPACITY, "tuple buffer full (%d + %d > %d): fpc_tuples_reserve", e->tuple_count, G, e->tuple_cap);
  if (game_id) HIPCHK(e, hipMemcpyAsync(e->d_tgame, game_id, (size_t)G * sizeof(int), hipMemcpyHostToDevice, e->stream));
  FPC_LAUNCH(k_collect_tuples, G, 64, e->stream, e->t, G, game_id ? (const int *)e->d_tgame : (const int *)nullptr, ply, e->d_tuples + e->tuple_count);
  HIPCHK(e, hipGetLastError());
  if (game_id) HIPCHK(e, hipStreamSynchronize(e->stream));     // the caller may reuse game_id
  e->tuple_count += G;
  return 0;
}

int fpc_tuples_set_z(fpc_engine *e, const int *game_id, const float *z_team0, const float *z_team1, int n) {
  if (!e || !game_id || !z_team0 || !z_team1 || n < 0) return fail(e, FPC_EINVAL, "bad argument");
  if (n > e->cfg.max_games) return fail(e, FPC_EINVAL, "more games than max_games in one fpc_tuples_set_z call");
  if (n == 0 || e->tuple_count == 0) return 0;
  USE_DEV(e);
  HIPCHK(e, hipMemcpyAsync(e->d_tgame, game_id, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->d_tz, z_team0, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->d_tz + e->cfg.max_games, z_team1, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream));
  FPC_LAUNCH(k_tuples_set_z, (e->tuple_count + 255) / 256, 256, e->stream, e->d_tuples, e->tuple_count, (const int *)e->d_tgame,
             (const float *)e->d_tz, (const float *)(e->d_tz + e->cfg.max_games), n);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return 0;
}

int fpc_tuples_count(fpc_engine *e) { return e ? e->tuple_count : FPC_EINVAL; }

int fpc_tuples_read(fpc_engine *e, fpc_tuple *host_out, int first, int n) {
  if (!e || !host_out || first < 0 || n < 0 || first + n > e->tuple_count) return fail(e, FPC_EINVAL, "bad tuple range");
  if (n == 0) return 0;
  USE_DEV(e);
  HIPCHK(e, hipMemcpyAsync(host_out, e->d_tuples + first, (size_t)n * sizeof(fpc_tuple), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// RCCL (backend "nccl" on ROCm), bound at run time: the copy PyTorch-ROCm already mapped into the
// process is preferred (one HIP runtime, one RCCL), else the system librccl.so.
#ifndef FPC_EMUL
namespace {
struct Rccl {
  typedef int (*get_id_t)(void *);
  typedef int (*allgather_t)(const void *, void *, size_t, int, void *, hipStream_t);
  typedef int (*destroy_t)(void *);
  typedef const char *(*errstr_t)(int);
  void *h = nullptr;
  destroy_t abort = nullptr;        // ncclCommAbort (optional)
  get_id_t get_id = nullptr;
  void *init_rank = nullptr;
  allgather_t allgather = nullptr;
  destroy_t destroy = nullptr;
  errstr_t errstr = nullptr;
  std::string err;
};
struct Id128 { char b[128]; };    // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value

extern "C++" Rccl &rccl() {
  static Rccl r;
  if (r.h || !r.err.empty()) return r;
  // FPC_RCCL_LIB: the host layer names the copy that belongs to the HIP runtime already in the process
  // (PyTorch-ROCm ships its own librccl.so next to its libamdhip64.so)
  if (const char *env = getenv("FPC_RCCL_LIB")) {
    r.h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) {                                   // an explicit choice: no silent substitute
      const char *de = dlerror();                 // (dlerror() clears the message when read: once per site)
      r.err = std::string("FPC_RCCL_LIB=") + env + " cannot be loaded: " + (de ? de : "");
      return r;
    }
  }
  const char *names[] = {"librccl.so", "librccl.so.1"};
  for (const char *n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char *n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) {
    const char *de = dlerror();
    r.err = std::string("librccl.so not found: ") + (de ? de : "");
    return r;
  }
  r.get_id = (Rccl::get_id_t)dlsym(r.h, "ncclGetUniqueId");
  r.init_rank = dlsym(r.h, "ncclCommInitRank");
  r.allgather = (Rccl::allgather_t)dlsym(r.h, "ncclAllGather");
  r.destroy = (Rccl::destroy_t)dlsym(r.h, "ncclCommDestroy");
  r.errstr = (Rccl::errstr_t)dlsym(r.h, "ncclGetErrorString");
  r.abort = (Rccl::destroy_t)dlsym(r.h, "ncclCommAbort");
  if (!r.get_id || !r.init_rank || !r.allgather || !r.destroy) r.err = "librccl.so lacks ncclGetUniqueId/CommInitRank/AllGather/CommDestroy";
  return r;
}
int comm_fail(fpc_engine *e, const char *what, int rc) {
  Rccl &r = rccl();
  return fail(e, FPC_ECOMM, "%s failed: ncclResult %d (%s)", what, rc, r.errstr ? r.errstr(rc) : "?");
}
}  // namespace
#endif

int fpc_comm_available(void) {
#ifdef FPC_EMUL
  return fail(nullptr, FPC_EUNSUPPORTED, "RCCL exists only in the gfx950 build");
#else
  Rccl &r = rccl();
  return r.err.empty() ? 0 : fail(nullptr, FPC_ECOMM, "%s", r.err.c_str());
#endif
}

int fpc_comm_unique_id(void *id128) {
  if (!id128) return FPC_EINVAL;
#ifdef FPC_EMUL
  return fail(nullptr, FPC_EUNSUPPORTED, "RCCL exists only in the gfx950 build");
#else
  Rccl &r = rccl();
  if (!r.err.empty()) return fail(nullptr, FPC_ECOMM, "%s", r.err.c_str());
  const int rc = r.get_id(id128);
  return rc ? comm_fail(nullptr, "ncclGetUniqueId", rc) : 0;
#endif
}

int fpc_comm_init(fpc_engine *e, const void *id128, int rank, int world) {
  if (!e || !id128 || world < 1 || rank < 0 || rank >= world) return fail(e, FPC_EINVAL, "bad argument");
#ifdef FPC_EMUL
  return fail(e, FPC_EUNSUPPORTED, "RCCL exists only in the gfx950 build");
#else
  USE_DEV(e);
  Rccl &r = rccl();
  if (!r.err.empty()) return fail(e, FPC_ECOMM, "%s", r.err.c_str());
  if (e->comm) return fail(e, FPC_ESTATE, "communicator already initialised");
  Id128 id;
  memcpy(&id, id128, sizeof(id));
  typedef int (*init_fn)(void **, int, Id128, int);
  const int rc = ((init_fn)r.init_rank)(&e->comm, world, id, rank);
  if (rc) { e->comm = nullptr; return comm_fail(e, "ncclCommInitRank", rc); }
  e->comm_rank = rank; e->comm_world = world;
  int rr;
  if (!e->d_gcounts || e->gcounts_world < world) {
    if ((rr = dalloc(e, &e->d_gcounts, 3 * (size_t)world + 4))) return rr;     // zero-filled: epoch 0 is never a fresh record
    e->gcounts_world = world;
  }
  // a new communicator starts at epoch 0 with every record cleared: a word left behind by an earlier communicator
  // must never pass for a fresh one
  e->comm_epoch = 0;
  if (hipMemset(e->d_gcounts, 0, (3 * (size_t)e->gcounts_world + 4) * sizeof(long long)) != hipSuccess) {
    (void)r.destroy(e->comm); e->comm = nullptr;      // local: nobody waits in a collective yet (tuples.init_comm MIN-reduces the verdicts)
    return fail(e, FPC_ENODEVICE, "clearing the exchange records failed");
  }
  e->gather_counts.assign(world, 0);
  return 0;
#endif
}

int fpc_comm_destroy(fpc_engine *e) {
  if (!e) return FPC_EINVAL;
#ifndef FPC_EMUL
  if (e->comm) {
    (void)hipSetDevice(e->cfg.device);
    (void)rccl().destroy(e->comm);
    e->comm = nullptr;
  }
#endif
  return 0;
}

// The episode-end exchange.  What it guarantees: NO LOCAL FAILURE CAN STRAND A PEER.  Every rank that
// enters goes through the same fixed sequence of collectives -- counts, status, payload -- and which of them run depends
// only on values every rank has agreed on, never on a local success:
//   1. counts round: (tuple count, buffer capacity, epoch) per rank.  A rank whose own HIP calls fail on the way in STILL
//      enters the collective; its slot then carries the previous exchange's epoch, which the others recognise;
//   2. status round, ALWAYS (one 8-byte all-gather per episode): 2 * epoch + 1 = "everything up to here worked on this
//      rank" (counts read back, every peer's record fresh, send / receive buffers big enough).  Anything else -- 2 * epoch
//      (failed) or a stale word (the status upload itself failed) -- makes EVERY rank leave before the payload
//      collective, the failing one with its own error, the others with FPC_ECOMM naming the rank.  The protocol stays in
//      lockstep, so the next exchange on the same communicator works (tests/test_comm_two_ranks_gpu.py);
//   3. payload round.
// Two windows cannot be closed by agreement: an error returned by an RCCL call itself, and a failure to read the status
// words back (the rank cannot know whether the others go on).  There the rank aborts its communicator (ncclCommAbort, where
// the library has it) so that peers blocked on it get an error instead of waiting, and returns; fpc_comm_init is needed again.
int fpc_allgather_tuples(fpc_engine *e, int *counts_out, int *total_out) {
  if (!e || !counts_out || !total_out) return fail(e, FPC_EINVAL, "bad argument");
#ifdef FPC_EMUL
  return fail(e, FPC_EUNSUPPORTED, "RCCL exists only in the gfx950 build");
#else
  if (!e->comm) return fail(e, FPC_ESTATE, "fpc_comm_init has not been called");
  USE_DEV(e);
  Rccl &r = rccl();
  const int W = e->comm_world;
  const long long epoch = ++e->comm_epoch;
  int local_rc = 0;
  std::string local_err;
  auto note = [&](int rc) { if (rc && !local_rc) { local_rc = rc; local_err = e->err; } };
  // a test can make one of this function's HIP calls fail (fpc_debug_comm_fault): the call is then NOT made
  auto faulted = [&](int point) -> bool {
    if (e->comm_fault != point) return false;
    e->comm_fault = 0;
    return true;
  };
  auto hip_ok = [&](hipError_t he, const char *what) -> bool {
    if (he == hipSuccess) return true;
    note(fail(e, FPC_ENODEVICE, "fpc_allgather_tuples: %s failed: %s", what, hipGetErrorString(he)));
    return false;
  };
  auto abort_comm = [&](int rc) -> int {      // leave between collectives without agreement: peers must not wait for this rank
    const std::string keep = e->err;
    if (r.abort) (void)r.abort(e->comm); else (void)r.destroy(e->comm);
    e->comm = nullptr;
    e->err = keep + " (communicator aborted: fpc_comm_init is needed again)";
    return rc;
  };
  long long *slot = e->d_gcounts + 3 * (size_t)W;       // this rank's counts record (3 words) ...
  long long *stslot = slot + 3;                          // ... and its status word: a slot of its own, so that what a failed upload
                                                         // leaves there is the PREVIOUS exchange's status word, never a fresh one
  // 1. counts round
  long long mine[3] = {e->tuple_count, std::min<long long>(e->tuple_cap, (long long)(e->gather_cap / (size_t)W)), epoch};
  hip_ok(faulted(1) ? hipErrorUnknown : hipMemcpyAsync(slot, mine, sizeof(mine), hipMemcpyHostToDevice, e->stream), "counts upload");
  int rc = r.allgather(slot, e->d_gcounts, 3, /*ncclInt64*/ 4, e->comm, e->stream);
  if (rc) return abort_comm(comm_fail(e, "ncclAllGather(counts)", rc));
  std::vector<long long> cnt3(3 * (size_t)W, 0), cnt(W, 0);
  if (hip_ok(faulted(2) ? hipErrorUnknown : hipMemcpyAsync(cnt3.data(), e->d_gcounts, cnt3.size() * sizeof(long long), hipMemcpyDeviceToHost, e->stream), "counts read-back"))
    hip_ok(hipStreamSynchronize(e->stream), "counts read-back");
  long long mx = 1;
  if (!local_rc) {
    for (int i = 0; i < W; ++i) {
      if (cnt3[3 * (size_t)i + 2] != epoch || cnt3[3 * (size_t)i] < 0) {
        note(fail(e, FPC_ECOMM, "rank %d sent no fresh counts record for this exchange (its upload failed)", i));
        break;
      }
      cnt[i] = cnt3[3 * (size_t)i];
      mx = std::max(mx, cnt[i]);
    }
  }
  // 1b. room for the padded payload on both sides (every rank knows mx, so each grows its own buffers)
  if (!local_rc && mx > e->tuple_cap) {
    fpc_tuple *bigger = nullptr;
    if (faulted(3)) note(fail(e, FPC_ENOMEM, "fpc_allgather_tuples: growing the tuple send buffer failed (injected)"));
    else note(dalloc(e, &bigger, (size_t)mx));
    if (!local_rc) {
      hipError_t he = hipSuccess;
      if (e->tuple_count) he = hipMemcpyAsync(bigger, e->d_tuples, (size_t)e->tuple_count * sizeof(fpc_tuple), hipMemcpyDeviceToDevice, e->stream);
      if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
      if (!hip_ok(he, "growing the tuple send buffer")) {
        (void)hipFree(bigger);
        e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), (void *)bigger));
      } else {
        if (e->d_tuples) { (void)hipFree(e->d_tuples); e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), (void *)e->d_tuples)); }
        e->d_tuples = bigger;
        e->tuple_cap = (int)mx;
      }
    }
  }
  const size_t want = (size_t)W * (size_t)mx;           // tuples in the receive buffer (size_t: world x capacity can pass 2^31)
  if (!local_rc && want > e->gather_cap) {
    if (e->d_gather) { (void)hipFree(e->d_gather); e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), (void *)e->d_gather)); e->d_gather = nullptr; }
    e->gather_cap = 0;
    note(dalloc(e, &e->d_gather, want));
    if (!local_rc) e->gather_cap = want;
  }
  // 2. status round -- always
  const long long good = 2 * epoch + 1;
  long long st = local_rc == 0 ? good : 2 * epoch;
  std::vector<long long> sts(W, 0);
  hip_ok(faulted(4) ? hipErrorUnknown : hipMemcpyAsync(stslot, &st, sizeof(st), hipMemcpyHostToDevice, e->stream), "status upload");   // failed: the slot keeps a stale word
  rc = r.allgather(stslot, e->d_gcounts, 1, /*ncclInt64*/ 4, e->comm, e->stream);
  if (rc) return abort_comm(comm_fail(e, "ncclAllGather(status)", rc));
  const int before = local_rc;
  if (hip_ok(faulted(5) ? hipErrorUnknown : hipMemcpyAsync(sts.data(), e->d_gcounts, (size_t)W * sizeof(long long), hipMemcpyDeviceToHost, e->stream), "status read-back"))
    hip_ok(hipStreamSynchronize(e->stream), "status read-back");
  if (local_rc && !before) { e->err = local_err; return abort_comm(local_rc); }     // said "good", cannot see what the others said
  if (local_rc) { e->err = local_err; return local_rc; }                              // the others saw this rank's word: nobody goes on
  for (int i = 0; i < W; ++i)
    if (sts[i] != good) return fail(e, FPC_ECOMM, "rank %d reported a failure before the payload collective; no rank entered it", i);
  // 3. payload, padded to the largest count: ONE collective per episode (latency-bound)
  e->gather_stride = (int)mx;
  rc = r.allgather(e->d_tuples, e->d_gather, (size_t)mx * sizeof(fpc_tuple), /*ncclUint8*/ 1, e->comm, e->stream);
  if (rc) return abort_comm(comm_fail(e, "ncclAllGather(tuples)", rc));
  HIPCHK(e, hipStreamSynchronize(e->stream));            // behind the last collective: nobody waits for this rank any more
  int total = 0;
  for (int i = 0; i < W; ++i) { e->gather_counts[i] = (int)cnt[i]; counts_out[i] = (int)cnt[i]; total += (int)cnt[i]; }
  *total_out = total;
  return 0;
#endif
}

// TEST HOOK: the next fpc_allgather_tuples on this engine treats one of its own HIP calls as failed --
// 1 counts upload, 2 counts read-back, 3 send-buffer growth, 4 status upload, 5 status read-back (0 = none).
int fpc_debug_comm_fault(fpc_engine *e, int point) {
  if (!e || point < 0 || point > 5) return fail(e, FPC_EINVAL, "bad argument");
  e->comm_fault = point;
  return 0;
}

int fpc_gathered_read(fpc_engine *e, fpc_tuple *host_out, int first, int n) {
  if (!e || !host_out || first < 0 || n < 0) return fail(e, FPC_EINVAL, "bad argument");
#ifdef FPC_EMUL
  return fail(e, FPC_EUNSUPPORTED, "RCCL exists only in the gfx950 build");
#else
  if (!e->d_gather) return fail(e, FPC_ESTATE, "fpc_allgather_tuples has not been called");
  USE_DEV(e);
  // logical index -> (rank, local index): ranks are stored max-padded
  int r = 0, base = 0, i = first, left = n;
  fpc_tuple *dst = host_out;
  while (left > 0) {
    while (r < e->comm_world && i >= base + e->gather_counts[r]) { base += e->gather_counts[r]; ++r; }
    if (r >= e->comm_world) return fail(e, FPC_EINVAL, "gathered tuple range out of bounds");
    const int loc = i - base, take = std::min(left, e->gather_counts[r] - loc);
    HIPCHK(e, hipMemcpyAsync(dst, e->d_gather + (size_t)r * e->gather_stride + loc, (size_t)take * sizeof(fpc_tuple), hipMemcpyDeviceToHost, e->stream));
    dst += take; i += take; left -= take;
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return 0;
#endif
}

int fpc_stats_get(fpc_engine *e, fpc_stats *out) {
  if (!e || !out) return FPC_EINVAL;
  *out = e->stats;
  return 0;
}
int fpc_stats_reset(fpc_engine *e) {
  if (!e) return FPC_EINVAL;
  e->stats = fpc_stats{};
  return 0;
}
int fpc_set_policy_mode(fpc_engine *e, int mode) {
  if (!e || (mode != FPC_POLICY_FULL && mode != FPC_POLICY_LEGAL)) return fail(e, FPC_EINVAL, "bad policy mode");
  e->policy_mode = mode;
  return 0;
}

int fpc_set_rules(fpc_engine *e, int rules) {
  if (!e || rules < 0 || rules > FPC_RULES_FIXED) return fail(e, FPC_EINVAL, "bad rule set");
  if (e->searching && rules != e->dc.rules) e->searching = false;     // a search in flight was started under the other rules
  e->dc.rules = rules;
#ifndef FPC_EMUL
  e->nn.dc.rules = rules;
#endif
  return 0;
}

int fpc_search_set_root_noise(fpc_engine *e, const float *gamma, int n_games, float eps) {
  if (!e) return FPC_EINVAL;
  USE_DEV(e);
  if (!gamma) { e->t.noise = nullptr; e->t.noise_eps = 0.f; e->noise_n = 0; return 0; }
  if (n_games < 1 || n_games > e->cfg.max_games || !(eps >= 0.f && eps <= 1.f)) return fail(e, FPC_EINVAL, "bad root-noise arguments");
  int r;
  if (!e->d_noise && (r = dalloc(e, &e->d_noise, (size_t)e->cfg.max_games * FPC_MAX_MOVES))) return r;
  HIPCHK(e, hipMemcpyAsync(e->d_noise, gamma, (size_t)n_games * FPC_MAX_MOVES * sizeof(float), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->t.noise = e->d_noise;
  e->t.noise_eps = eps;
  e->noise_n = n_games;
  return 0;
}

int fpc_set_timing(fpc_engine *e, int enabled) {
  if (!e) return FPC_EINVAL;
  e->timing = enabled != 0;
  return 0;
}
const char *fpc_nn_kernel(fpc_engine *e) {
#ifdef FPC_EMUL
  (void)e;
  return "";
#else
  if (!e || !e->nn.loaded) return "";
  return e->nn.use_towerw ? "k_towerw" : e->nn.use_tower ? (e->nn.tower_compact && e->nn.tower_waves == 8 && e->dc.R == 14 ? "k_towerc" : "k_tower") : "k_conv3x3";
#endif
}
void *fpc_stream(fpc_engine *e) { return e ? (void *)e->stream : nullptr; }

#ifdef FPC_TREE_STAMPS
// diagnostic builds only (tools/tree_stamps.py): the s_memtime stamps the tree kernels left for game FPC_TREE_STAMPS
int fpc_debug_tree_stamps(unsigned long long *out32) {
  return hipMemcpyFromSymbol(out32, HIP_SYMBOL(fpc::g_tree_stamps), 32 * sizeof(unsigned long long)) == hipSuccess ? 0 : FPC_ENODEVICE;
}
#endif

}  // extern "C"
