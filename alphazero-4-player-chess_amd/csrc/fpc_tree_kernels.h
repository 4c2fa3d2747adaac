// fpc_tree_kernels.h -- CDNA4 kernels for the tree side of MCTS.search: one 64-lane wavefront per
// concurrent game (block = 1 wave, grid = games), board state staged in LDS.
//
//   k_board_ops      batched Board::GetLegalMoves / GetGameResult / TakeAction / encode / mask
//   k_attack_maps    GetAttackedSquaresPlayers / GetAttackedSquaresTeams / IsAttackedByPlayer (wrapper.cpp:201-206)
//   k_select         Node::ChooseLeaf: PUCT descent (wave argmax), lazy leaf-board materialisation,
//                    GetGameResult + GetLegalMoves on the leaf, terminal backup        (node.cpp:19-78)
//   k_encode         Board::GetEncodedStates incl. the batch-wide rot90                 (board.cpp:305-356)
//   k_softmax_partials  per-chunk softmax statistics of a logits row (external evaluators; the internal
//                    policy Linear's k_fc_reduce leaves the same records)                (mcts.py:67)
//   k_expand         ParseActionspace / mask / renormalise from those statistics, BackpropagateNodes,
//                    ExpandNodes                                                        (mcts.py:67-89)
//   k_expand_select  k_expand of simulation step s + k_select of step s+1 in one launch
//
// Reference semantics (file:line relative to /root/reference/src/cpp) are restated per function.
// Nothing here is translated from the reference: its board is a pointer-rich mailbox + std::vector
// piece lists walked serially; here one wave works on a 288-byte LDS image, (piece,direction)
// slots are walked by separate lanes, every pseudo-legal move is legality-tested by its own lane on
// a *virtual* post-move board (no make/undo), and the piece-list reorderings that the reference's
// make/undo loops cause (observable through GetGameResult) are applied in closed form.
#pragma once
#include "fpc_platform.h"
#include "../../include/fpc_engine.h"

namespace fpc {

enum { PAWN = 0, KNIGHT = 1, BISHOP = 2, ROOK = 3, QUEEN = 4, KING = 5 };

struct DevCfg {
  int R, INV, RR, A_ch, A;
  int rules;         // FPC_RULES_* bits (include/fpc_engine.h); 0 = strict reference semantics
  int rcpR;          // 65536 / R + 1: q / R == (q * rcpR) >> 16 exactly for 0 <= q < 4681 (R <= 14)
};
__host__ __device__ __forceinline__ DevCfg make_devcfg(int R, int INV, int rules) {
  return DevCfg{R, INV, R * R, 8 * R + 8, (8 * R + 8) * R * R, rules, 65536 / R + 1};
}
// square index -> row (the integer division by the runtime board size costs ~40 VALU instructions)
__host__ __device__ __forceinline__ int row_of(const DevCfg &c, int q) { return (q * c.rcpR) >> 16; }

// error bits accumulated per game / per board (host maps them to fpc_status)
enum { ERR_SELECT = 1, ERR_POLICY = 2, ERR_CAP_MOVES = 4, ERR_CAP_NODES = 8, ERR_MOVE = 16, ERR_CAP_BOARDS = 32 };

// ---- per-game tree storage (SoA, one contiguous region per game) -------------------------------
struct Tree {
  int *N;            // visit_count                     node.h:70
  double *W;         // value_sum                       node.h:76
  float *P;          // prior (f32 -> double on use)    node.h:75, mcts.py:87
  uint16_t *mv;      // move_made as flat index         node.h:74
  int *parent;       // -1 for root                     node.h:73
  int *child0;       // first child (children contiguous, ascending flat), -1 = not expanded  node.h:78
  uint16_t *nch;
  int *bslot;        // board-pool slot of this node's state, -1 = not materialised yet
  fpc_board *boards; // board pool
  int node_cap, board_cap;
  // per game scalars
  int *nnodes, *nboards, *alive, *sims_done, *err;
  int *leaf_node;    // leaf chosen this step (-1: none)
  int *leaf_slot;    // board-pool slot of that leaf's state (-1: none) -- what k_encode reads
  int *leaf_turn;
  int *leaf_node_nx, *leaf_slot_nx, *leaf_turn_nx;   // the same three for the NEXT step (k_expand_select writes them while
                                                     // other games' expansions still read this step's; the host swaps)
  int *nlegal;
  uint16_t *legal;   // [G][FPC_MAX_MOVES] ascending unique flat indices of the leaf's legal moves
  int *path;         // [G][path_cap] root..leaf node ids of this step's descent (k_select -> k_expand's backup)
  int *path_len;
  int path_cap;
  // root Dirichlet noise (N4; off when null): gamma draws [G][FPC_MAX_MOVES], one per root child in ascending flat order
  const float *noise;
  float noise_eps;
};

// ---- LDS image of one wave --------------------------------------------------------------------
constexpr int GEN_SCR = 14;             // per generator slot: a ray has at most 13 targets; slot 7 of a king: 1 step + 2 castlings
struct __attribute__((aligned(16))) WaveLds {
  fpc_board b;                         // 288 B
  // pseudo-legal moves in the reference's generation order.  mw packs what the later phases read:
  //   bits 0-7 to, 8-15 captured piece byte, 17 promotion (the reference emits 4 variants), 20-23 the
  //   mover's piece-list position (generation order key), 24-31 castling: the rook's square (it hops to
  //   from + (to-from)/2), else FPC_NO_SQ
  uint32_t mw[FPC_MAX_MOVES];
  uint8_t mfrom[FPC_MAX_MOVES];
  uint16_t lflat[FPC_MAX_MOVES];       // flat index of legal moves, reference order
  uint16_t lsorted[FPC_MAX_MOVES];     // ascending
  uint8_t lidx[FPC_MAX_MOVES];         // pseudo-move index of k-th legal move
  float pri[FPC_MAX_MOVES];
  uint32_t scr[64][2][GEN_SCR];        // generation scratch: lane x generator slot x targets (same packing as mw)
  uint16_t poff[FPC_MAX_PL + 1];       // first pseudo-move of each piece-list entry
  uint8_t l1pos[FPC_MAX_PL];           // own list: position after the GetGameResult reordering
  uint8_t newlist[3][FPC_MAX_PL];      // reordered lists (own, enemy a, enemy b) before they are copied back
  alignas(4) uint8_t ent[200];         // square -> 16*list + entry for the three lists a move can touch, else 0xFF (cleared 4 bytes per lane)
  int lk[2][48];                       // last move touching each list entry (GetGameResult's loop, GetLegalMoves' loop)
  int M, nlegal, first_legal, result, errbits;
  float scal_f;                        // wave-uniform scalar broadcast slot
};
__device__ __forceinline__ int mv_to(uint32_t w) { return (int)(w & 255u); }
__device__ __forceinline__ uint8_t mv_cap(uint32_t w) { return (uint8_t)(w >> 8); }
__device__ __forceinline__ int mv_promo(uint32_t w) { return (int)((w >> 17) & 1u); }
__device__ __forceinline__ int mv_piece(uint32_t w) { return (int)((w >> 20) & 15u); }
__device__ __forceinline__ int mv_rook(uint32_t w) { return (int)(w >> 24); }

__device__ __forceinline__ bool present(uint8_t p) { return (p & 0x80) != 0; }
__device__ __forceinline__ int colour_of(uint8_t p) { return (p >> 5) & 3; }
__device__ __forceinline__ int type_of(uint8_t p) { return (p >> 2) & 7; }
__device__ __forceinline__ int team_of_colour(int c) { return c & 1; }   // RED/YELLOW 0, BLUE/GREEN 1 (engine/board.h:64-67)
__device__ __forceinline__ int team_of(uint8_t p) { return (p >> 5) & 1; }

// engine/board.h:647-654.  Branch-free on purpose (bitwise & |): these run per lane in divergent code, where
// every short-circuit turns into an exec-mask save/restore and a branch.
__device__ __forceinline__ bool legal_loc(const DevCfg &c, int row, int col) {
  const int mx = c.R - 1;
  const bool in = ((unsigned)row <= (unsigned)mx) & ((unsigned)col <= (unsigned)mx);
  const bool cc = (col < c.INV) | (col > mx - c.INV);
  const bool rr = (row < c.INV) | (row > mx - c.INV);
  return in & !(cc & rr);
}
__host__ __device__ __forceinline__ bool in_array(const DevCfg &c, int row, int col) {
  return ((unsigned)row < (unsigned)c.R) & ((unsigned)col < (unsigned)c.R);
}

// move.cpp:13-20, :84-104: (from,to) -> action plane
__host__ __device__ __forceinline__ int move_plane(const DevCfg &c, int from, int to) {
  const int fr = row_of(c, from), tr = row_of(c, to);
  const int dy = tr - fr, dx = (to - tr * c.R) - (from - fr * c.R);
  const int ay = dy < 0 ? -dy : dy, ax = dx < 0 ? -dx : dx;
  if (dx == 0 || dy == 0 || ax == ay) {
    const int sx = (dx > 0) - (dx < 0), sy = (dy > 0) - (dy < 0);
    // queen_move_offsets (dx,dy): {0,-1},{-1,-1},{-1,0},{-1,1},{0,1},{1,1},{1,0},{1,-1}
    const int key = (sx + 1) * 3 + (sy + 1);
    int dir;
    switch (key) {
      case 0: dir = 1; break; case 1: dir = 2; break; case 2: dir = 3; break;
      case 3: dir = 0; break; case 5: dir = 4; break;
      case 6: dir = 7; break; case 7: dir = 6; break; default: dir = 5; break;
    }
    const int dist = ax > ay ? ax : ay;
    return dir * (c.R - 1) + dist - 1;
  }
  // knight_move_offsets (dx,dy): {-2,-1},{-2,1},{-1,-2},{-1,2},{1,-2},{1,2},{2,-1},{2,1}
  const int k = (dx == -2 ? 0 : dx == -1 ? 2 : dx == 1 ? 4 : 6) + (dy > 0 ? 1 : 0);
  return 8 * (c.R - 1) + k;
}

// Move(flat): move.cpp:39-61.  returns `to` (FPC_NO_SQ when off the array / unaddressable plane)
__host__ __device__ __forceinline__ int flat_to(const DevCfg &c, int flat, int *from_out) {
  const int plane = flat / c.RR, from = flat % c.RR;
  *from_out = from;
  const int nq = c.R - 1;
  int dx, dy;
  if (plane < 8 * nq) {
    const int dir = plane / nq, dist = plane % nq + 1;
    const int qdx[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    const int qdy[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
    dx = qdx[dir] * dist; dy = qdy[dir] * dist;
  } else {
    const int k = plane - 8 * nq;
    if (k >= 8) return FPC_NO_SQ;
    const int ndx[8] = {-2, -2, -1, -1, 1, 1, 2, 2};
    const int ndy[8] = {-1, 1, -2, 2, -2, 2, -1, 1};
    dx = ndx[k]; dy = ndy[k];
  }
  const int row = from / c.R + dy, col = from % c.R + dx;
  if (!in_array(c, row, col)) return FPC_NO_SQ;
  return row * c.R + col;
}

// input plane of a piece seen from the side to move (board.cpp:322-336): strict = 6*rel + type - 1 with
// -1 wrapping to 23 (quirk Q7); FPC_RULES_PLANES = 6*rel + type
__device__ __forceinline__ int piece_plane(uint8_t p, int turn, int rules) {
  int plane = 6 * ((colour_of(p) - turn) & 3) + type_of(p);
  if (!(rules & FPC_RULES_PLANES)) { plane -= 1; if (plane < 0) plane += 24; }
  return plane;
}

// torch.rot90(x, k, (-2,-1)) index map: out[i][j] = in[src]   (board.cpp:252-255)
__device__ __forceinline__ int rot90_src(int R, int k, int i, int j) {
  k &= 3;
  return k == 0 ? i * R + j : k == 1 ? j * R + (R - 1 - i) : k == 2 ? (R - 1 - i) * R + (R - 1 - j) : (R - 1 - j) * R + i;
}

// ---- wave helpers -------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// diagnostic builds only (-DFPC_TREE_STAMPS=<block>; tools/tree_stamps.py): s_memtime at the phase boundaries of the
// tree kernels for ONE game, read back through fpc_debug_tree_stamps.  The product build compiles none of it.
#ifdef FPC_TREE_STAMPS
__device__ unsigned long long g_tree_stamps[32];
#define FPC_TS(I) do { if ((int)blockIdx.x == FPC_TREE_STAMPS && lane_id() == 0) g_tree_stamps[I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FPC_TS(I) do {} while (0)
#endif

__device__ __forceinline__ void lds_load_board(WaveLds *s, const fpc_board *g) {
  const uint32_t *src = reinterpret_cast<const uint32_t *>(g);
  uint32_t *dst = reinterpret_cast<uint32_t *>(&s->b);
  const int l = lane_id();
  dst[l] = src[l];
  if (l < 8) dst[64 + l] = src[64 + l];
  __syncthreads();
}
__device__ __forceinline__ void lds_store_board(const WaveLds *s, fpc_board *g) {
  __syncthreads();
  const uint32_t *src = reinterpret_cast<const uint32_t *>(&s->b);
  uint32_t *dst = reinterpret_cast<uint32_t *>(g);
  const int l = lane_id();
  dst[l] = src[l];
  if (l < 8) dst[64 + l] = src[64 + l];
}

__device__ __forceinline__ bool promotes(const DevCfg &c, int colour, int tr, int tc) {   // engine/board.cpp:58-76
  switch (colour) {
    case 0: return tr == c.R / 4;
    case 1: return tc == 3 * c.R / 4;
    case 2: return tr == 3 * c.R / 4;
    default: return tc == c.R / 4;
  }
}

// Piece-list primitive of RemovePiece / SetPiece (engine/board.cpp:977-1014), wave-cooperative: lanes
// 0..15 hold one entry each, the first entry equal to `sq` is erased (the tail closes up) and `add` (>= 0)
// is appended.  Returns the new length; *found tells whether `sq` was there.  All lanes must call.
__device__ __forceinline__ int wave_list_erase_append(uint8_t *list, int len, int sq, int add, bool *found) {
  const int lane = lane_id();
  const int v = (lane < len && lane < FPC_MAX_PL) ? list[lane] : -1;
  const int nxt = __shfl(v, lane + 1 < 64 ? lane + 1 : lane);
  const unsigned long long hit = __ballot(v == sq);
  __syncthreads();                                   // every entry is in a register before any is rewritten
  const int idx = hit ? (int)__ffsll((long long)hit) - 1 : -1;
  int n = len;
  if (idx >= 0) {
    if (lane >= idx && lane + 1 < len) list[lane] = (uint8_t)nxt;
    n = len - 1;
  }
  if (add >= 0 && n < FPC_MAX_PL) { if (lane == 0) list[n] = (uint8_t)add; ++n; }
  *found = idx >= 0;
  __syncthreads();
  return n;
}

// MakeMove for a tree/self-play move, which carries only (from,to): capture whatever stands on
// `to`, no promotion, no rook hop, no rights update (engine/board.cpp:1028-1096 with a
// Move(flat)/Move(plane,from) argument).  false: "piece missing".
// With FPC_RULES_FULL_MOVES (the non-strict rule set) the move is executed the way the
// reference's own generator describes it (engine/board.cpp:58-88, :256-302, :313-466): a pawn that
// reaches its promotion line becomes a queen, a two-square king move hops the rook, king moves clear the
// mover's castling rights and a rook leaving its home square clears that side's.
// Wave-cooperative and wave-uniform: every lane derives the same scalars from the LDS board, the piece
// lists are edited one entry per lane, lane 0 writes the scalars.  All lanes must call.
__device__ inline bool make_move_wave(fpc_board *b, int from, int to, const DevCfg &c) {
  const int lane = lane_id();
  if (to == FPC_NO_SQ) return false;
  uint8_t piece = b->sq[from];
  const uint8_t cap = b->sq[to];
  bool found;
  __syncthreads();
  if (present(cap)) {  // RemovePiece(to)
    const int cc = colour_of(cap);
    const int n = wave_list_erase_append(b->pl[cc], b->plen[cc], to, -1, &found);
    if (lane == 0) {
      b->plen[cc] = (uint8_t)n;
      b->sq[to] = 0;
      if (type_of(cap) == KING) b->king[cc] = FPC_NO_SQ;
    }
    __syncthreads();
  }
  if (!present(piece)) return false;
  const int pc = colour_of(piece);
  int rook_from = FPC_NO_SQ, rook_to = FPC_NO_SQ;
  uint8_t rights = b->castle[pc];
  if (c.rules & FPC_RULES_FULL_MOVES) {
    const int R = c.R, fr = row_of(c, from), fc = from - fr * R, tr = row_of(c, to), tc = to - tr * R;
    const int ty = type_of(piece);
    if (ty == PAWN) {
      if (promotes(c, pc, tr, tc)) piece = (uint8_t)(0x80 | (pc << 5) | (QUEEN << 2));
    } else if (ty == KING) {
      const int dr = tr - fr, dc = tc - fc;
      if ((dr == 0 && (dc == 2 || dc == -2)) || (dc == 0 && (dr == 2 || dr == -2))) {   // castling: the rook hops next to the king
        const int ur = dr / 2, uc = dc / 2;
        // kingside: rook 3 squares from the king, queenside: 4 (engine/board.cpp:343-465)
        for (int d = 3; d <= 4; ++d) {
          const int rr = fr + ur * d, rc = fc + uc * d;
          if (in_array(c, rr, rc)) {
            const uint8_t rk = b->sq[rr * R + rc];
            if (present(rk) && type_of(rk) == ROOK && colour_of(rk) == pc) { rook_from = rr * R + rc; rook_to = (fr + ur) * R + fc + uc; break; }
            if (present(rk)) break;
          }
        }
      }
      rights = 0;
    } else if (ty == ROOK && rights) {
      int ks, qs;      // rook home squares, engine/board.cpp:256-289
      switch (pc) {
        case 0: ks = (R - 1) * R + (R - 4); qs = (R - 1) * R + c.INV; break;
        case 1: ks = (R - 4) * R; qs = c.INV * R; break;
        case 2: ks = c.INV; qs = R - 4; break;
        default: ks = c.INV * R + (R - 1); qs = (R - 4) * R + (R - 1); break;
      }
      if (from == ks) rights &= (uint8_t)~1u;
      if (from == qs) rights &= (uint8_t)~2u;
    }
  }
  const uint8_t rook_piece = rook_from != FPC_NO_SQ ? b->sq[rook_from] : (uint8_t)0;
  // RemovePiece(from) + SetPiece(to, piece)
  int n = wave_list_erase_append(b->pl[pc], b->plen[pc], from, to, &found);
  if (rook_from != FPC_NO_SQ) {   // RemovePiece(rook_from) + SetPiece(rook_to): the rook's entry follows the king's
    const int n2 = wave_list_erase_append(b->pl[pc], n, rook_from, rook_to, &found);
    if (!found) { if (lane == 0 && n > 0) b->pl[pc][n - 1] = (uint8_t)rook_to; } else n = n2;
  }
  if (lane == 0) {
    b->plen[pc] = (uint8_t)n;
    b->sq[from] = 0;
    b->sq[to] = piece;
    if (type_of(piece) == KING) b->king[pc] = (uint8_t)to;
    if (c.rules & FPC_RULES_FULL_MOVES) b->castle[pc] = rights;
    if (rook_from != FPC_NO_SQ) { b->sq[rook_to] = rook_piece; b->sq[rook_from] = 0; }
    b->turn = (uint8_t)((b->turn + 1) & 3);  // GetNextPlayer, engine/board.cpp:1299-1313
  }
  __syncthreads();
  return true;
}

// IsAttackedByTeam(team, ksq) on the position obtained from `b` by moving the piece on `from` to
// `to` -- and, for castling, the rook on `from2` to `to2` -- (virtual make; FPC_NO_SQ squares -> the
// position itself).  engine/board.cpp:606-787.
// The function has no side effects, so the reference's probe ORDER is not observable; what costs time
// on a single wave is the LDS round trip of every dependent probe (~100 cycles each).  All 20 leaper
// probes (8 knight, 4 pawn, 8 king squares) and the first ATT_PRE squares of the 8 rays are therefore
// loaded back to back (out-of-range probes read the king's own square and are ignored), evaluated from
// registers, and only a ray still open after ATT_PRE empty squares continues with dependent loads.
constexpr int ATT_PRE = 3;
__device__ __attribute__((noinline)) bool attacked_virtual(const fpc_board *b, const DevCfg &c, int from, int to, uint8_t mover,
                                        int from2, int to2, uint8_t mover2, int ksq, int team) {
  const int R = c.R;
  const int kr = row_of(c, ksq), kc = ksq - kr * R;
  // content of square q after the virtual move: a chain of plain selects (lowest priority first), no branches
  auto vsq = [&](int q, uint32_t raw) -> uint8_t {
    uint32_t v = raw;
    v = q == to2 ? (uint32_t)mover2 : v;
    v = q == to ? (uint32_t)mover : v;
    v = ((q == from) | (q == from2)) ? 0u : v;
    return (uint8_t)v;
  };
#define FPC_VSQ(q, raw) vsq((q), (raw))
  // a piece byte is 1 c c t t t 0 0 (present, colour, type): present + team + type in one masked compare
  const uint8_t tkey = (uint8_t)(0x80 | (team << 5));
  int lq[20];
  uint8_t lraw[20];
  // knights: all 8 offsets regardless of board size (:676-694)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dr = (k & 4) ? ((k & 2) ? 1 : -1) : ((k & 2) ? 2 : -2);
    const int dc = (k & 4) ? ((k & 1) ? 2 : -2) : ((k & 1) ? 1 : -1);
    const int r = kr + dr, cc = kc + dc;
    lq[k] = legal_loc(c, r, cc) ? r * R + cc : -1;
  }
  // pawns (:697-750): array bounds only
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (k >> 1) ? kr + 1 : kr - 1, cc = (k & 1) ? kc + 1 : kc - 1;
    lq[8 + k] = in_array(c, r, cc) ? r * R + cc : -1;
  }
  // kings (:753-772)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kk = k < 4 ? k : k + 1;
    const int r = kr + kk / 3 - 1, cc = kc + kk % 3 - 1;
    lq[12 + k] = legal_loc(c, r, cc) ? r * R + cc : -1;
  }
  // rays: rooks & queens end at the ARRAY edge, not at the cut corners (:632);
  // bishops & queens are bounded by IsLegalLocation (:658)
  int rq[8][ATT_PRE];
  uint8_t rraw[8][ATT_PRE];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const int ri = d < 4 ? (d == 0 ? -1 : d == 1 ? 1 : 0) : ((d & 2) ? 1 : -1);
    const int ci = d < 4 ? (d == 2 ? -1 : d == 3 ? 1 : 0) : ((d & 1) ? 1 : -1);
#pragma unroll
    for (int k = 0; k < ATT_PRE; ++k) {
      const int r = kr + ri * (k + 1), cc = kc + ci * (k + 1);
      const bool in = d < 4 ? in_array(c, r, cc) : legal_loc(c, r, cc);
      rq[d][k] = in ? r * R + cc : -1;
    }
  }
#pragma unroll
  for (int k = 0; k < 20; ++k) lraw[k] = b->sq[lq[k] < 0 ? ksq : lq[k]];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int k = 0; k < ATT_PRE; ++k) rraw[d][k] = b->sq[rq[d][k] < 0 ? ksq : rq[d][k]];
  bool hit = false;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint8_t p = FPC_VSQ(lq[k], lraw[k]);
    hit |= (lq[k] >= 0) & ((p & 0xBC) == (tkey | (KNIGHT << 2)));
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint8_t p = FPC_VSQ(lq[8 + k], lraw[8 + k]);
    // the pawn of colour `col` attacks diagonally forward: red (0) upwards, so it sits one row below (pr = 1), ...
    const int pr = k >> 1, pc = k & 1, col = colour_of(p);
    const bool att = ((col == 0) & (pr != 0)) | ((col == 1) & (pc == 0)) | ((col == 2) & (pr == 0)) | ((col == 3) & (pc != 0));
    hit |= (lq[8 + k] >= 0) & ((p & 0xBC) == (tkey | (PAWN << 2))) & att;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint8_t p = FPC_VSQ(lq[12 + k], lraw[12 + k]);
    hit |= (lq[12 + k] >= 0) & ((p & 0xBC) == (tkey | (KING << 2)));
  }
  unsigned open_rays = 0;     // bit d: ray d saw ATT_PRE empty in-range squares
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const uint8_t slider = (uint8_t)(tkey | ((d < 4 ? ROOK : BISHOP) << 2)), queen = (uint8_t)(tkey | (QUEEN << 2));
    bool open = true;
#pragma unroll
    for (int k = 0; k < ATT_PRE; ++k) {
      const uint8_t p = FPC_VSQ(rq[d][k], rraw[d][k]);
      const bool in = rq[d][k] >= 0, occ = present(p);
      hit |= open & in & (((p & 0xBC) == slider) | ((p & 0xBC) == queen));
      open = open & in & !occ;
    }
    open_rays |= open ? 1u << d : 0u;
  }
  if (open_rays && !hit) {   // more than ATT_PRE empty squares in some direction: walk on with dependent loads
    for (int d = 0; d < 8; ++d) {
      if (!((open_rays >> d) & 1u)) continue;
      const int slider = d < 4 ? ROOK : BISHOP;
      const int ri = d < 4 ? (d == 0 ? -1 : d == 1 ? 1 : 0) : ((d & 2) ? 1 : -1);
      const int ci = d < 4 ? (d == 2 ? -1 : d == 3 ? 1 : 0) : ((d & 1) ? 1 : -1);
      int r = kr + ri * (ATT_PRE + 1), cc = kc + ci * (ATT_PRE + 1);
      while (d < 4 ? in_array(c, r, cc) : legal_loc(c, r, cc)) {
        const int q = r * R + cc;
        const uint8_t p = FPC_VSQ(q, b->sq[q]);
        if (present(p)) {
          hit |= team_of(p) == team && (type_of(p) == slider || type_of(p) == QUEEN);
          break;
        }
        r += ri; cc += ci;
      }
    }
  }
#undef FPC_VSQ
  return hit;
}

// ---- king-safety tables: IsKingSafeAfterMove (board.cpp:59-68) for ALL pseudo-legal moves of a position at once ----
// attacked_virtual above answers "is square K attacked after this move" from scratch: 44 probes and ~1 000 VALU
// instructions per move, on every lane.  But for the moves of one position almost everything is shared: the king
// stands on the same square K for every move that is not a king move, and such a move changes the board on two squares
// only.  So the wave looks at K ONCE:
//   * the enemy leapers (knight / pawn / king, IsAttackedByTeam's probe squares, engine/board.cpp:676-772) that attack K
//     now: their number and, if it is one, its square -- a move of another piece removes such an attack only by
//     capturing that piece;
//   * per ray d from K (rook-type rays end at the ARRAY edge, bishop-type rays at the first illegal location, exactly
//     as in engine/board.cpp:632 / :658): the distance of the first and of the second piece on it and whether each is an
//     enemy slider of the ray's type -- a move changes what K sees along a ray only by interposing on / capturing along
//     that ray (the mover becomes the first piece: the mover's side never attacks its own king) or by taking the FIRST
//     piece off it (the second one becomes visible).
// A lane then decides its move with a few compares (ks_move_leaves_king_attacked).  King moves -- K itself changes --
// are decided by eight lanes each, one per ray / leaper offset (ks_king_moves).  Castling (never generated through the
// reference's Python boundary) and positions judged for a player of the other team than the mover keep the
// generic test.  Every decision equals attacked_virtual's: same squares, same bounds, same piece tests.
__device__ __forceinline__ void ray_dir(int d, int *ri, int *ci) {            // attacked_virtual's ray numbering
  *ri = d < 4 ? (d == 0 ? -1 : d == 1 ? 1 : 0) : ((d & 2) ? 1 : -1);
  *ci = d < 4 ? (d == 2 ? -1 : d == 3 ? 1 : 0) : ((d & 1) ? 1 : -1);
}
// ray (0..7, or -1) and distance of square (r, c) as seen from (kr, kc); branch-free
__device__ __forceinline__ int ray_of(int kr, int kc, int r, int c, int *dist) {
  const int dr = r - kr, dc = c - kc;
  const int ar = dr < 0 ? -dr : dr, ac = dc < 0 ? -dc : dc;
  int d = -1;
  d = ((dc == 0) & (dr != 0)) ? (dr < 0 ? 0 : 1) : d;
  d = ((dr == 0) & (dc != 0)) ? (dc < 0 ? 2 : 3) : d;
  d = ((ar == ac) & (ar != 0)) ? 4 + (dr > 0 ? 2 : 0) + (dc > 0 ? 1 : 0) : d;
  *dist = ar > ac ? ar : ac;
  return d;
}
__device__ __forceinline__ bool enemy_slider(uint8_t p, int d, int team) {
  const uint8_t tkey = (uint8_t)(0x80 | (team << 5));
  return ((p & 0xBC) == (uint8_t)(tkey | ((d < 4 ? ROOK : BISHOP) << 2))) | ((p & 0xBC) == (uint8_t)(tkey | (QUEEN << 2)));
}
// leaper probe k (0..7 knight, 8..11 pawn, 12..19 king) of square (kr, kc): the probed square or -1, as attacked_virtual
__device__ __forceinline__ int leaper_square(const DevCfg &c, int kr, int kc, int k) {
  int r, cc;
  bool in;
  if (k < 8) {
    const int dr = (k & 4) ? ((k & 2) ? 1 : -1) : ((k & 2) ? 2 : -2);
    const int dc = (k & 4) ? ((k & 1) ? 2 : -2) : ((k & 1) ? 1 : -1);
    r = kr + dr; cc = kc + dc; in = legal_loc(c, r, cc);
  } else if (k < 12) {
    const int j = k - 8;
    r = (j >> 1) ? kr + 1 : kr - 1; cc = (j & 1) ? kc + 1 : kc - 1; in = in_array(c, r, cc);
  } else {
    const int j = k - 12, kk = j < 4 ? j : j + 1;
    r = kr + kk / 3 - 1; cc = kc + kk % 3 - 1; in = legal_loc(c, r, cc);
  }
  return in ? r * c.R + cc : -1;
}
// does piece byte p on leaper probe k attack the probed-from square for `team`?  (engine/board.cpp:676-772)
__device__ __forceinline__ bool leaper_attacks(uint8_t p, int k, int team) {
  const uint8_t tkey = (uint8_t)(0x80 | (team << 5));
  if (k < 8) return (p & 0xBC) == (uint8_t)(tkey | (KNIGHT << 2));
  if (k >= 12) return (p & 0xBC) == (uint8_t)(tkey | (KING << 2));
  const int j = k - 8, pr = j >> 1, pc = j & 1, col = colour_of(p);
  const bool att = ((col == 0) & (pr != 0)) | ((col == 1) & (pc == 0)) | ((col == 2) & (pr == 0)) | ((col == 3) & (pc != 0));
  return ((p & 0xBC) == (uint8_t)(tkey | (PAWN << 2))) & att;
}
struct KingSafety {
  int kr, kc;          // the king's square K
  uint32_t ray;        // THIS lane's copy of ray (lane & 7): d1 | d2 << 4 | len << 8 | a1 << 12 | a2 << 13  (fetch another ray's with __shfl)
  int abase;           // bit d: the first piece on ray d is an enemy slider of that ray's type (wave-uniform)
  int nla, sqla;       // enemy leapers attacking K now; the square of the only one when nla == 1 (wave-uniform)
};
// All 64 lanes call.  K must be a square of the board.
__device__ __forceinline__ KingSafety ks_build(const fpc_board *b, const DevCfg &c, int K, int team) {
  const int lane = lane_id(), R = c.R;
  KingSafety ks;
  ks.kr = row_of(c, K); ks.kc = K - ks.kr * R;
  // rays: lane (d, k) looks at distances k + 1 and k + 9 of ray d
  {
    const int d = lane >> 3, k = lane & 7;
    int ri, ci;
    ray_dir(d, &ri, &ci);
    const int r1 = ks.kr + ri * (k + 1), c1 = ks.kc + ci * (k + 1), r2 = ks.kr + ri * (k + 9), c2 = ks.kc + ci * (k + 9);
    const bool in1 = d < 4 ? in_array(c, r1, c1) : legal_loc(c, r1, c1);
    const bool in2 = (k < 5) & (d < 4 ? in_array(c, r2, c2) : legal_loc(c, r2, c2));
    const uint8_t p1 = b->sq[in1 ? r1 * R + c1 : K], p2 = b->sq[in2 ? r2 * R + c2 : K];
    const unsigned long long Bin1 = __ballot(in1), Bin2 = __ballot(in2);
    const unsigned long long Boc1 = __ballot(in1 & present(p1)), Boc2 = __ballot(in2 & present(p2));
    // every lane assembles ray dd = lane & 7 (the eight copies are identical; lane dd < 8 is the one others fetch)
    const int dd = lane & 7;
    const uint32_t in13 = (uint32_t)((Bin1 >> (8 * dd)) & 0xFFull) | ((uint32_t)((Bin2 >> (8 * dd)) & 0x1Full) << 8);
    const uint32_t oc13 = (uint32_t)((Boc1 >> (8 * dd)) & 0xFFull) | ((uint32_t)((Boc2 >> (8 * dd)) & 0x1Full) << 8);
    const int len = __builtin_ctz(~in13);                       // squares up to the first one that is out of range (<= 13)
    const uint32_t oc = oc13 & ((1u << len) - 1u);
    const int d1 = oc ? __builtin_ctz(oc) + 1 : 0;
    const uint32_t oc2 = oc & (oc - 1u);
    const int d2 = oc2 ? __builtin_ctz(oc2) + 1 : 0;
    int rdi, cdi;
    ray_dir(dd, &rdi, &cdi);
    const uint8_t q1 = b->sq[d1 ? (ks.kr + rdi * d1) * R + ks.kc + cdi * d1 : K];
    const uint8_t q2 = b->sq[d2 ? (ks.kr + rdi * d2) * R + ks.kc + cdi * d2 : K];
    const bool a1 = (d1 != 0) & enemy_slider(q1, dd, team), a2 = (d2 != 0) & enemy_slider(q2, dd, team);
    ks.ray = (uint32_t)d1 | ((uint32_t)d2 << 4) | ((uint32_t)len << 8) | (a1 ? 1u << 12 : 0u) | (a2 ? 1u << 13 : 0u);
    ks.abase = (int)(__ballot(a1) & 0xFFull);                   // lanes 0..7 hold rays 0..7
  }
  // leapers: lane k < 20 owns probe k
  {
    const int q = lane < 20 ? leaper_square(c, ks.kr, ks.kc, lane) : -1;
    const uint8_t p = b->sq[q < 0 ? K : q];
    const bool att = (q >= 0) && leaper_attacks(p, lane, team);
    const unsigned long long B = __ballot(att);
    ks.nla = __popcll(B);
    ks.sqla = wave_read(q, B ? (int)__ffsll((long long)B) - 1 : 0);
  }
  return ks;
}
// Is K attacked after the move f -> t of a piece of K's own side that is not K itself (no rook hop)?  All lanes call
// (the ray records travel by __shfl); lanes without a move pass any squares and ignore the answer.
__device__ __forceinline__ bool ks_move_leaves_king_attacked(const KingSafety &ks, const DevCfg &c, int f, int t) {
  const int R = c.R;
  const int fr = row_of(c, f), fc = f - fr * R, tr = row_of(c, t), tc = t - tr * R;
  int dt, df;
  const int rt = ray_of(ks.kr, ks.kc, tr, tc, &dt), rf = ray_of(ks.kr, ks.kc, fr, fc, &df);
  const uint32_t pt = __shfl(ks.ray, rt < 0 ? 0 : rt), pf = __shfl(ks.ray, rf < 0 ? 0 : rf);
  int att = ks.abase;
  {   // the mover lands on ray rt in front of (or on) its first piece: it is the first piece now
    const int d1 = (int)(pt & 15u), len = (int)((pt >> 8) & 15u);
    const bool blocks = (rt >= 0) & (dt <= len) & ((d1 == 0) | (dt <= d1));
    att = blocks ? att & ~(1 << (rt < 0 ? 0 : rt)) : att;
  }
  {   // the mover WAS the first piece of ray rf and leaves the ray: the second piece shows
    const int d1 = (int)(pf & 15u);
    const bool opens = (rf >= 0) & (d1 != 0) & (df == d1) & (rt != rf);
    att = (opens & (((pf >> 13) & 1u) != 0u)) ? att | (1 << (rf < 0 ? 0 : rf)) : att;
  }
  const bool leaper = (ks.nla >= 2) | ((ks.nla == 1) & (t != ks.sqla));
  return (att != 0) | leaper;
}
// Up to eight king moves f -> T_j at once: lane (j, d) walks ray d from T_j (with f emptied) and probes leaper offsets
// d of the knight / king patterns and d < 4 of the pawn pattern.  tsq: T_j of slot j = lane >> 3, or -1 (empty slot).
// Returns the ballot of "attacks T_j" over all lanes: slot j's answer is bits 8 j .. 8 j + 7 != 0.  All lanes call.
__device__ __forceinline__ unsigned long long ks_king_moves(const fpc_board *b, const DevCfg &c, int f, int tsq, int team) {
  const int lane = lane_id(), R = c.R, d = lane & 7;
  bool hit = false;
  if (tsq >= 0) {
    const int tr = row_of(c, tsq), tc = tsq - tr * R;
    // leapers: three independent probes
    const int qn = leaper_square(c, tr, tc, d), qk = leaper_square(c, tr, tc, 12 + d), qp = d < 4 ? leaper_square(c, tr, tc, 8 + d) : -1;
    const uint8_t pn = b->sq[qn < 0 ? tsq : qn], pk = b->sq[qk < 0 ? tsq : qk], pp = b->sq[qp < 0 ? tsq : qp];
    hit |= (qn >= 0) & (qn != f) && leaper_attacks(pn, d, team);
    hit |= (qk >= 0) & (qk != f) && leaper_attacks(pk, 12 + d, team);
    hit |= (qp >= 0) & (qp != f) && leaper_attacks(pp, 8 + (d & 3), team);
    // the ray: first piece on the board with f emptied (the king now stands on tsq itself)
    int ri, ci;
    ray_dir(d, &ri, &ci);
    int r = tr + ri, cc = tc + ci;
    while (d < 4 ? in_array(c, r, cc) : legal_loc(c, r, cc)) {
      const int q = r * R + cc;
      const uint8_t p = q == f ? (uint8_t)0 : b->sq[q];
      if (present(p)) { hit |= enemy_slider(p, d, team); break; }
      r += ri; cc += ci;
    }
  }
  return __ballot(hit);
}

// Generator slot `d` (0..7) of a piece, as data: a start square, a step and what a target square may
// hold.  The reference's generators -- pawn {fwd1, fwd2, capture-, capture+} (engine/board.cpp:97-177),
// knight (:179-207, loop bound invalid_area = quirk Q8), bishop (:240-254), rook (:256-302), queen =
// bishop then rook (:304-311), king 8 steps (:313-341) -- all reduce to "step (ir,ic) up to n times".
enum { GEN_QUIET = 1, GEN_CAPT = 2, GEN_PUSH2 = 4, GEN_PAWN = 8 };
struct GenSlot {
  int ir, ic;   // step
  int n;        // steps left (0: the slot is finished / does not exist)
  int mode;     // GEN_QUIET: empty targets are moves; GEN_CAPT: an enemy-occupied target is a move;
                // GEN_PUSH2: pawn double step (the 1st square must be a legal, empty location and is not a
                // move; the 2nd is only checked against the array bounds, Q15); GEN_PAWN: promotion line
};
__device__ __forceinline__ GenSlot gen_slot(const DevCfg &c, int type, int colour, int fr, int fc, int d) {
  GenSlot g{0, 0, 0, 0};
  const int R = c.R;
  if (type == PAWN) {
    if (d >= 4) return g;
    int dr = 0, dc = 0;
    bool not_moved;
    switch (colour) {
      case 0: dr = -1; not_moved = fr == R - 2; break;
      case 1: dc = 1; not_moved = fc == 1; break;
      case 2: dr = 1; not_moved = fr == 1; break;
      default: dc = -1; not_moved = fc == R - 2; break;
    }
    g.ir = dr; g.ic = dc;
    if (d == 0) { g.n = 1; g.mode = GEN_QUIET | GEN_PAWN; }
    else if (d == 1) { if (not_moved) { g.n = 2; g.mode = GEN_QUIET | GEN_PUSH2 | GEN_PAWN; } }
    else {
      const int sd = d == 2 ? -1 : 1;
      if ((colour & 1) == 0) g.ic += sd; else g.ir += sd;
      g.n = 1; g.mode = GEN_CAPT | GEN_PAWN;
    }
  } else if (type == KNIGHT) {
    const int per = 2 * (c.INV - 1);           // 0, 2 or 4 (invalid_area 1..3)
    if (d >= 2 * per) return g;
    const int sh = per == 4 ? 2 : 1;
    const int prs = d >> sh, rem = d & (per - 1), adr = 1 + (rem >> 1), pcs = rem & 1;
    const int adc = adr == 1 ? 2 : 1;
    g.ir = prs ? adr : -adr; g.ic = pcs ? adc : -adc;
    g.n = 1; g.mode = GEN_QUIET | GEN_CAPT;
  } else if (type == KING) {
    const int k = d < 4 ? d : d + 1;
    g.ir = k / 3 - 1; g.ic = k % 3 - 1;
    g.n = 1; g.mode = GEN_QUIET | GEN_CAPT;
  } else if (type == BISHOP || type == ROOK || type == QUEEN) {   // AddMovesFromIncrMovement2, engine/board.cpp:209-238
    const bool diag = type == BISHOP || (type == QUEEN && d < 4);
    const int rd = type == QUEEN ? d - 4 : d;
    if (diag) {
      if (d >= 4) return g;
      g.ir = (d & 2) ? 1 : -1; g.ic = (d & 1) ? 1 : -1;
    } else {
      // rook generator order (:291-301): (0,-1), (-1,0), (0,+1), (+1,0)
      if (rd < 0 || rd >= 4) return g;
      const int incr = (rd & 2) ? 1 : -1;
      if (rd & 1) { g.ir = incr; g.ic = 0; } else { g.ir = 0; g.ic = incr; }
    }
    g.n = R; g.mode = GEN_QUIET | GEN_CAPT;
  }
  return g;
}
// Castling (engine/board.cpp:343-465), generated after the king's eight steps, queenside first:
// rights bit set, a same-team rook on the expected square, empty squares between, and neither the
// king's square nor the first square it crosses attacked.  emit(to, rook_from).  The move travels as
// a plain two-square king move through the tree (Move(flat) drops the rook hop); only
// GetLegalMoves/GetGameResult execute the hop (virtually, for the legality test) and reorder lists.
template <class F>
__device__ inline void walk_castle(const fpc_board *b, const DevCfg &c, int from, F &&emit) {
  const uint8_t king = b->sq[from];
  const int colour = colour_of(king), team = team_of(king);
  const uint8_t cur = b->castle[colour];
  if (!(cur & 3)) return;
  const int R = c.R, fr = from / R, fc = from % R;
  for (int is_k = 0; is_k < 2; ++is_k) {
    if (!(is_k ? (cur & 1) : (cur & 2))) continue;
    int ur = 0, uc = 0;
    switch (colour) {
      case 0: uc = is_k ? 1 : -1; break;
      case 1: ur = is_k ? 1 : -1; break;
      case 2: uc = is_k ? -1 : 1; break;
      default: ur = is_k ? -1 : 1; break;
    }
    const int nb = is_k ? 2 : 3;
    const int rr = fr + ur * (nb + 1), rc = fc + uc * (nb + 1);
    if (!in_array(c, rr, rc)) continue;                 // reference reads out of bounds here
    const uint8_t rook = b->sq[rr * R + rc];
    if (!present(rook) || type_of(rook) != ROOK || team_of(rook) != team) continue;
    bool blocked = false;
    for (int k = 1; k <= nb; ++k) blocked |= present(b->sq[(fr + ur * k) * R + fc + uc * k]);
    if (blocked) continue;
    const int b0 = (fr + ur) * R + fc + uc;
    if (attacked_virtual(b, c, FPC_NO_SQ, FPC_NO_SQ, 0, FPC_NO_SQ, FPC_NO_SQ, 0, b0, team ^ 1)) continue;
    if (attacked_virtual(b, c, FPC_NO_SQ, FPC_NO_SQ, 0, FPC_NO_SQ, FPC_NO_SQ, 0, from, team ^ 1)) continue;
    emit((fr + 2 * ur) * R + fc + 2 * uc, rr * R + rc);
  }
}

// Wave-cooperative movegen + legality + (optionally) GetGameResult / GetLegalMoves side effects.
//   do_result: run Board::GetGameResult(player) (engine/board.cpp:891-939) -> s->result, and apply
//              the piece-list reordering of its make/undo loop (stops at the first legal move).
//   do_legal : run fpchess::Board::GetLegalMoves (board.cpp:94-118) -> s->lflat/lidx/lsorted,
//              s->nlegal, and apply the reordering of its full make/undo loop.  Skipped when
//              do_result found a terminal position (ChooseLeaf returns before the mask is built).
// `player` < 0 -> side to move.  All lanes of the wave must call this (uniform control flow).
__device__ inline void wave_position_ops(WaveLds *s, const DevCfg &c, bool do_result, bool do_legal, int player) {
  const int lane = lane_id();
  fpc_board *b = &s->b;
  const int turn = b->turn;
  if (player < 0) player = turn;
  const int nown = b->plen[turn];
  const bool gen = b->king[turn] != FPC_NO_SQ;   // GetPseudoLegalMoves2 returns 0 without own king (:852-856)
  const bool player_has_king = b->king[player] != FPC_NO_SQ;

  // ---- generation, one walk: lane = (piece p, generator slots d0, d0+1).  Both slots advance in the same
  //      loop (two independent LDS round trips per iteration) and park their targets in the lane's scratch
  //      rows; the wave scan of the counts then gives every lane its place in the reference's generation order.
  const int p = lane >> 2, d0 = (lane & 3) * 2;
  int cnt0 = 0, cnt1 = 0;
  int from = FPC_NO_SQ;
  uint32_t *scr0 = s->scr[lane][0], *scr1 = s->scr[lane][1];
  if (gen && p < nown) {
    from = b->pl[turn][p];
    const uint8_t piece = b->sq[from];
    const int type = type_of(piece), colour = colour_of(piece), team = team_of(piece);
    const int R = c.R, fr = row_of(c, from), fc = from - fr * R;
    GenSlot g0 = gen_slot(c, type, colour, fr, fc, d0), g1 = gen_slot(c, type, colour, fr, fc, d0 + 1);
    int r0 = fr, c0 = fc, r1 = fr, c1 = fc;
    auto step = [&](GenSlot &g, int &tr, int &tc, uint32_t *scr, int &cnt) {
      if (g.n <= 0) return;
      tr += g.ir; tc += g.ic;
      const bool second = (g.mode & GEN_PUSH2) && g.n == 1;
      if (!(second ? in_array(c, tr, tc) : legal_loc(c, tr, tc))) { g.n = 0; return; }
      const int to = tr * R + tc;
      const uint8_t cap = b->sq[to];
      bool emit;
      if (!present(cap)) {
        emit = (g.mode & GEN_QUIET) && !((g.mode & GEN_PUSH2) && g.n == 2);
        g.n = (g.mode & GEN_QUIET) ? g.n - 1 : 0;
      } else {
        emit = (g.mode & GEN_CAPT) && team_of(cap) != team;
        g.n = 0;
      }
      if (emit && cnt < GEN_SCR) {
        const bool promo = (g.mode & GEN_PAWN) && promotes(c, colour, tr, tc);
        scr[cnt++] = (uint32_t)to | ((uint32_t)cap << 8) | (promo ? 1u << 17 : 0u) | ((uint32_t)FPC_NO_SQ << 24);
      }
    };
    while (g0.n > 0 || g1.n > 0) {
      step(g0, r0, c0, scr0, cnt0);
      step(g1, r1, c1, scr1, cnt1);
    }
    if (d0 == 6 && type == KING)
      walk_castle(b, c, from, [&](int to, int rook_from) {
        if (cnt1 < GEN_SCR) scr1[cnt1++] = (uint32_t)to | ((uint32_t)rook_from << 24);
      });
  }
  FPC_TS(7);
  // exclusive wave scan of (cnt0+cnt1) in lane order == reference generation order
  int incl = cnt0 + cnt1;
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const int total = wave_read(incl, 63);
  const int base = incl - (cnt0 + cnt1);
  if ((lane & 3) == 0 && p <= FPC_MAX_PL - 1) s->poff[p] = (uint16_t)(base < FPC_MAX_MOVES ? base : FPC_MAX_MOVES);
  if (lane == 0) {
    s->poff[FPC_MAX_PL] = (uint16_t)(total < FPC_MAX_MOVES ? total : FPC_MAX_MOVES);
    s->errbits = total > FPC_MAX_MOVES ? ERR_CAP_MOVES : 0;
    if (nown > FPC_MAX_PL) s->errbits |= ERR_CAP_MOVES;
  }
  const int M = total < FPC_MAX_MOVES ? total : FPC_MAX_MOVES;
  {  // scratch rows -> the dense move list
    const uint32_t tag = (uint32_t)p << 20;
    const int kmax = cnt0 > cnt1 ? cnt0 : cnt1;
    for (int k = 0; k < kmax; ++k) {
      if (k < cnt0 && base + k < FPC_MAX_MOVES) { s->mw[base + k] = scr0[k] | tag; s->mfrom[base + k] = (uint8_t)from; }
      if (k < cnt1 && base + cnt0 + k < FPC_MAX_MOVES) { s->mw[base + cnt0 + k] = scr1[k] | tag; s->mfrom[base + cnt0 + k] = (uint8_t)from; }
    }
  }
  // square -> list entry map and the "last move touching this entry" slots of the reordering below
  if (lane < 50) reinterpret_cast<uint32_t *>(s->ent)[lane] = 0xFFFFFFFFu;
  if (lane < 48) { s->lk[0][lane] = -1; s->lk[1][lane] = -1; }
  __syncthreads();

  // ---- legality (IsKingSafeAfterMove, board.cpp:59-68): lane i decides move i.  The player's king square is looked
  //      at once for the whole position (ks_build); a move of another piece is then a few compares, king moves take
  //      eight lanes each (ks_king_moves); castling and "judged for the other team" keep attacked_virtual.
  FPC_TS(8);
  const int enemy = team_of_colour(player) ^ 1;
  const int K0 = b->king[player];
  const bool tables = (K0 != FPC_NO_SQ) & (team_of_colour(turn) == team_of_colour(player));   // wave-uniform
  KingSafety ks{};
  if (tables) ks = ks_build(b, c, K0, enemy);
  FPC_TS(9);
  int nlegal = 0, first = -1;
  for (int base_i = 0; base_i < M; base_i += 64) {
    const int i = base_i + lane;
    bool legal = false;
    int f = 0, t = 0;
    bool king_move = false, generic = false;
    uint32_t w = 0;
    uint8_t mover = 0;
    if (i < M) {
      w = s->mw[i];
      f = s->mfrom[i]; t = mv_to(w);
      mover = b->sq[f];
      king_move = (type_of(mover) == KING) & (colour_of(mover) == player);
      generic = !tables | (mv_rook(w) != FPC_NO_SQ);
      if (K0 == FPC_NO_SQ) { legal = true; generic = false; king_move = false; }   // no king to protect (the reference's ksq test never runs)
    }
    if (tables) {
      // moves of the other pieces: the tables
      const bool att = ks_move_leaves_king_attacked(ks, c, f, t);
      if (i < M && !king_move && !generic) legal = !att;
      // king moves (without the rook hop): slot j = rank among this chunk's king moves, eight lanes per slot
      const unsigned long long kb = __ballot(i < M && king_move && !generic);
      if (kb) {                                                  // wave-uniform
        const int myslot = __popcll(kb & ((1ull << lane) - 1ull));
        if ((kb >> lane) & 1ull) { if (myslot < 8) s->lk[0][myslot] = t | (f << 8); }   // lk is cleared again below
        __syncthreads();
        const int j = lane >> 3, nk = __popcll(kb);
        const int ent = j < nk && j < 8 ? s->lk[0][j] : -1;
        const unsigned long long hb = ks_king_moves(b, c, ent < 0 ? 0 : (ent >> 8), ent < 0 ? -1 : (ent & 255), enemy);
        __syncthreads();
        if ((kb >> lane) & 1ull) {
          if (myslot < 8) legal = ((hb >> (8 * myslot)) & 0xFFull) == 0ull;
          else generic = true;                                   // more than eight king moves in one chunk: cannot happen (8 steps), kept total
        }
        if (lane < 8) s->lk[0][lane] = -1;
        __syncthreads();
      }
    }
    if (__ballot(generic)) {                                     // wave-uniform guard around the out-of-line generic test
      if (generic) {
        int ksq = K0;
        if (king_move) ksq = t;
        else if (ksq == t) ksq = FPC_NO_SQ;                      // the player's king itself was captured
        const int rf = mv_rook(w);                               // castling: the rook hops next to the king
        const int rt = rf == FPC_NO_SQ ? FPC_NO_SQ : f + (t - f) / 2;
        legal = ksq == FPC_NO_SQ ? true : !attacked_virtual(b, c, f, t, mover, rf, rt, rf == FPC_NO_SQ ? (uint8_t)0 : b->sq[rf], ksq, enemy);
      }
    }
    const unsigned long long bal = __ballot(legal);
    if (legal) {
      const int k = nlegal + __popcll(bal & ((1ull << lane) - 1ull));
      s->lidx[k] = (uint8_t)i;
      s->lflat[k] = (uint16_t)(move_plane(c, f, t) * c.RR + f);
    }
    if (first < 0 && bal) first = base_i + (int)__ffsll((long long)bal) - 1;
    nlegal += __popcll(bal);
  }
  __syncthreads();

  FPC_TS(10);
  // ---- GetGameResult (engine/board.cpp:891-939)
  int result = FPC_IN_PROGRESS;
  if (do_result) {
    if (!player_has_king) {
      result = team_of_colour(player) == 0 ? FPC_WIN_BG : FPC_WIN_RY;
    } else if (nlegal > 0) {
      const uint8_t cap = mv_cap(s->mw[first]);          // only the FIRST legal move is inspected (Q13)
      if (present(cap) && type_of(cap) == KING) result = team_of(cap) == 0 ? FPC_WIN_BG : FPC_WIN_RY;
    } else {
      const bool chk = attacked_virtual(b, c, FPC_NO_SQ, FPC_NO_SQ, 0, FPC_NO_SQ, FPC_NO_SQ, 0, b->king[player], enemy);
      result = !chk ? FPC_STALEMATE : (team_of_colour(player) == 0 ? FPC_WIN_BG : FPC_WIN_RY);
    }
  }
  const bool run_legal = do_legal && result == FPC_IN_PROGRESS;

  // ---- piece-list reorderings caused by the reference's make/undo loops, in closed form.
  // Every make/undo pair moves the mover's entry -- then, for castling, the rook's entry -- and the
  // captured piece's entry to the END of their lists (engine/board.cpp:977-1014,1028-1160).
  // GetGameResult does that for moves 0..upto (it returns at the first legal move), GetLegalMoves
  // afterwards for ALL pseudo-legal moves, regenerated in the then-current list order.  Net effect on
  // every list: entries no move touches keep their order in front; touched entries follow, ordered by
  // the LAST move that touches them (key = position of that move in the loop's generation order).
  // Lanes 0-15 own the side-to-move entries, lanes 16-31 / 32-47 the entries of the two enemy colours.
  {
    const bool phase1 = do_result && player_has_king;
    const int upto = nlegal > 0 ? first : M - 1;
    const unsigned long long lower = (1ull << lane) - 1ull;
    const int grp = lane >> 4, e = lane & 15;                  // grp 0: own list; 1, 2: enemy colours turn+1, turn+3
    const int lcol = grp == 0 ? turn : (grp == 1 ? ((turn + 1) & 3) : ((turn + 3) & 3));
    const bool my_lane = grp <= 2 && e < b->plen[lcol] && e < FPC_MAX_PL;
    const int mysq = my_lane ? b->pl[lcol][e] : -1;
    const unsigned long long gmask = 0xFFFFull << (grp * 16);
    auto rank_in_group = [&](int key, bool touched) {          // touched entries of my list with a smaller key
      int rank = 0;
      for (int k = 0; k < 16; ++k) {
        const int ok = __shfl(key, (lane & 48) + k);
        rank += (touched && ok >= 0 && ok < key) ? 1 : 0;
      }
      return rank;
    };
    // Which entry a move touches is looked up through the square -> entry map, and every move posts its
    // key to the entries it touches with an LDS atomic max (keys grow along the loop, so the maximum is
    // the LAST move touching the entry): one step per move instead of every entry scanning all moves.
    if (my_lane) s->ent[mysq] = (uint8_t)lane;
    __syncthreads();
    auto post = [&](int *lk, int i, uint32_t w, int key) {
      atomicMax(&lk[mv_piece(w)], key);                          // the mover
      const int rook = mv_rook(w);
      if (rook != FPC_NO_SQ) {                                   // the rook is re-appended AFTER the king
        const int en = s->ent[rook];
        if (en < 16) atomicMax(&lk[en], key + 1);
      }
      if (present(mv_cap(w))) {                                  // the captured piece (always an enemy entry)
        const int en = s->ent[mv_to(w)];
        if (en >= 16 && en < 48) atomicMax(&lk[en], key);
      }
    };
    // GetGameResult's loop: key = move index
    if (phase1)
      for (int i = lane; i <= upto; i += 64) post(s->lk[0], i, s->mw[i], 2 * i);
    __syncthreads();
    const int lk1 = (phase1 && my_lane) ? s->lk[0][lane] : -1;
    const bool t1 = my_lane && lk1 >= 0;
    const unsigned long long T1 = __ballot(t1), L = __ballot(my_lane);
    const int r1 = rank_in_group(lk1, t1);
    const unsigned long long U1 = L & ~T1 & gmask;
    const int pos1 = t1 ? __popcll(U1) + r1 : __popcll(U1 & lower);
    if (grp == 0 && my_lane) s->l1pos[e] = (uint8_t)(phase1 ? pos1 : e);
    __syncthreads();
    // GetLegalMoves' loop: key = (post-GetGameResult position of the mover, its per-piece move index)
    if (run_legal)
      for (int i = lane; i < M; i += 64) {
        const uint32_t w = s->mw[i];
        const int pc = mv_piece(w);
        post(s->lk[1], i, w, 2 * ((int)s->l1pos[pc] * 256 + (i - (int)s->poff[pc])));
      }
    __syncthreads();
    const int lk2 = (run_legal && my_lane) ? s->lk[1][lane] : -1;
    const bool t2 = my_lane && lk2 >= 0;
    const unsigned long long T2 = __ballot(t2);
    const int r2 = rank_in_group(lk2, t2);
    const unsigned long long U2 = L & ~T2 & gmask;
    const int pos2 = t2 ? __popcll(U2) + r2 : __popcll(U2 & lower);
    const int newpos = run_legal ? pos2 : pos1;
    if ((run_legal || phase1) && my_lane) s->newlist[grp][newpos] = (uint8_t)mysq;
    __syncthreads();
    if ((run_legal || phase1) && my_lane) b->pl[lcol][e] = s->newlist[grp][e];
    if (lane == 0) { s->M = M; s->nlegal = run_legal ? nlegal : 0; s->first_legal = first; s->result = result; }
  }
  __syncthreads();

  FPC_TS(11);
  // ---- ascending flat order of the legal set (children are created in torch.nonzero order,
  //      mcts.py:84-87): rank sort, promotions are already collapsed to one entry
  if (run_legal) {
    for (int j = lane; j < nlegal; j += 64) {
      const uint16_t v = s->lflat[j];
      int rank = 0;
      for (int k = 0; k < nlegal; ++k) rank += s->lflat[k] < v;
      s->lsorted[rank] = v;
    }
  }
  __syncthreads();
}

// ================================================================================================
// k_board_ops: the batched Board API (one wave per position)
// ================================================================================================
enum { OP_LEGAL = 1, OP_RESULT = 2, OP_TAKE = 4 };

__global__ void __launch_bounds__(64) k_board_ops(DevCfg c, fpc_board *boards, int n, int ops, const int *player,
                                                  const int *flat, fpc_move *moves, int *counts, int *results,
                                                  int *err) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= n) return;
  const int lane = lane_id();
  lds_load_board(&s, &boards[g]);
  int e = 0;
  if (ops & OP_TAKE) {
    int from;
    const int to = flat_to(c, flat[g], &from);
    if (!make_move_wave(&s.b, from, to, c)) e |= ERR_MOVE;
  }
  if (ops & (OP_LEGAL | OP_RESULT)) {
    wave_position_ops(&s, c, (ops & OP_RESULT) != 0, (ops & OP_LEGAL) != 0, player ? player[g] : -1);
    e |= s.errbits;
    if (ops & OP_RESULT) { if (lane == 0) results[g] = s.result; }
    if (ops & OP_LEGAL) {
      const int nl = s.nlegal;
      if (lane == 0) counts[g] = nl;
      for (int k = lane; k < nl; k += 64) {
        const int i = s.lidx[k];
        fpc_move m;
        const uint32_t w = s.mw[i];
        m.from = s.mfrom[i]; m.to = (uint8_t)mv_to(w); m.capture = mv_cap(w); m.promo = (uint8_t)mv_promo(w);
        m.flat = s.lflat[k]; m.pad = 0;
        moves[(size_t)g * FPC_MAX_MOVES + k] = m;
      }
    }
  }
  lds_store_board(&s, &boards[g]);
  if (lane == 0) err[g] = e;
}

// ================================================================================================
// k_attack_maps: the attacked-square queries of the binding surface (wrapper.cpp:201-206), one wave per position:
//   maps 0..3  fpchess::Board::IsAttackedByPlayer(location, colour) (src/cpp/board.cpp:142-210) -- its OWN probe set, not
//              the engine's: pawns / knights / kings on the array (BoardLocation::Present(), engine/board.h:194-201), and
//              eight rays that end at the ARRAY edge, i.e. run through the cut corners; the first piece on a ray attacks
//              if it is a bishop (diagonals), rook (lines) or queen -- of whichever colour it is, which is the colour's bit;
//   maps 4..5  chess::Board::IsAttackedByTeam(team, location) (engine/board.cpp:606-787) = attacked_virtual on the position.
// Every row x column of the array is a location, cut corners included (the double loops of board.cpp:120-140, :212-232).
// out: [n][6][RR] bytes, 1 = attacked.
// ================================================================================================
__device__ inline uint32_t attacked_by_players(const fpc_board *b, const DevCfg &c, int sq) {
  const int R = c.R, lr = row_of(c, sq), lc = sq - lr * R;
  uint32_t m = 0;
  for (int k = 0; k < 8; ++k) {                       // the eight neighbours: pawns (:152-161) and kings (:200-207)
    const int kk = k < 4 ? k : k + 1;
    const int dr = kk / 3 - 1, dc = kk % 3 - 1;
    const int r = lr + dr, cc = lc + dc;
    if (!in_array(c, r, cc)) continue;
    const uint8_t p = b->sq[r * R + cc];
    if (!present(p)) continue;
    const int col = colour_of(p), t = type_of(p);
    if (t == KING) m |= 1u << col;
    // PawnAttacks(pawn, colour, location), engine/board.cpp:583-604: (row_diff, col_diff) = location - pawn = (-dr, -dc)
    const bool patt = (col == 0 && dr == 1 && dc != 0) || (col == 1 && dc == -1 && dr != 0) || (col == 2 && dr == -1 && dc != 0) ||
                      (col == 3 && dc == 1 && dr != 0);
    if (t == PAWN && patt) m |= 1u << col;
  }
  for (int k = 0; k < 8; ++k) {                       // knights (:164-171)
    const int dr = (k & 4) ? ((k & 2) ? 1 : -1) : ((k & 2) ? 2 : -2);
    const int dc = (k & 4) ? ((k & 1) ? 2 : -2) : ((k & 1) ? 1 : -1);
    const int r = lr + dr, cc = lc + dc;
    if (!in_array(c, r, cc)) continue;
    const uint8_t p = b->sq[r * R + cc];
    if (present(p) && type_of(p) == KNIGHT) m |= 1u << colour_of(p);
  }
  for (int k = 0; k < 8; ++k) {                       // sliders (:174-197): the first piece on each ray
    const int kk = k < 4 ? k : k + 1;
    const int dr = kk / 3 - 1, dc = kk % 3 - 1;
    const bool diag = dr != 0 && dc != 0;
    int r = lr + dr, cc = lc + dc;
    while (in_array(c, r, cc)) {
      const uint8_t p = b->sq[r * R + cc];
      if (present(p)) {
        const int t = type_of(p);
        if (t == QUEEN || (t == BISHOP && diag) || (t == ROOK && !diag)) m |= 1u << colour_of(p);
        break;
      }
      r += dr; cc += dc;
    }
  }
  return m;
}

__global__ void __launch_bounds__(64) k_attack_maps(DevCfg c, const fpc_board *boards, int n, uint8_t *out) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= n) return;
  lds_load_board(&s, &boards[g]);
  uint8_t *o = out + (size_t)g * 6 * c.RR;
  for (int sq = lane_id(); sq < c.RR; sq += 64) {
    const uint32_t pm = attacked_by_players(&s.b, c, sq);
    for (int col = 0; col < 4; ++col) o[col * c.RR + sq] = (uint8_t)((pm >> col) & 1u);
    for (int team = 0; team < 2; ++team)
      o[(4 + team) * c.RR + sq] = attacked_virtual(&s.b, c, FPC_NO_SQ, FPC_NO_SQ, 0, FPC_NO_SQ, FPC_NO_SQ, 0, sq, team) ? 1 : 0;
  }
}

// ================================================================================================
// k_encode: Board::GetEncodedStates (board.cpp:305-356).  plane = 6*((colour-turn)&3) + type - 1,
// -1 wrapping to 23 (Q7); whole batch rotated by the turn of the FIRST leaf (Q6).
// mode 0: f32 NCHW [G,24,R,R]   (reference layout; external evaluators, parity tests)
// mode 1: 16-bit NHWC on the zero-bordered (R+2)x(R+2) grid, 32 channels (24 + 8 zero) -- the
//         input image of the internal MFMA conv stack.  one16 = bit pattern of 1.0 in the NN dtype.
// ================================================================================================
__device__ __forceinline__ int first_leaf_turn(const int *leaf_node, const int *leaf_turn, int G) {
  const int lane = lane_id();
  int found = -1;
  for (int base = 0; base < G && found < 0; base += 64) {
    const int g = base + lane;
    const bool has = g < G && leaf_node[g] >= 0;
    const unsigned long long bal = __ballot(has);
    if (bal) found = base + (int)__ffsll((long long)bal) - 1;
  }
  return found < 0 ? 0 : leaf_turn[found];
}

__global__ void __launch_bounds__(64) k_encode(DevCfg c, const fpc_board *boards, int board_stride,
                                               const int *slot_of, const int *leaf_turn, int G, int mode,
                                               float *out_f32, uint16_t *out_nhwc, uint16_t one16,
                                               int fixed_rot /* <0: rotation from first live leaf */) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int lane = lane_id();
  const int slot = slot_of ? slot_of[g] : 0;   // for the search: leaf board slot (or -1 dead)
  const int R = c.R, RR = c.RR;
  const bool live = slot >= 0;
  const fpc_board *b = live ? &boards[(size_t)g * board_stride + slot] : nullptr;
  const int turn = live ? b->turn : 0;
  // batch-wide rotation by the first state's turn (Q6); FPC_RULES_ROTATION: every sample by its own turn
  const int k = (c.rules & FPC_RULES_ROTATION) ? turn : (fixed_rot >= 0 ? fixed_rot : first_leaf_turn(slot_of, leaf_turn, G));
  if (mode == 0) {
    float *o = out_f32 + (size_t)g * 24 * RR;
    for (int pos = lane; pos < RR; pos += 64) {
      int plane = -1;
      if (live) {
        const uint8_t p = b->sq[rot90_src(R, k, pos / R, pos % R)];
        if (present(p)) plane = piece_plane(p, turn, c.rules);
      }
      for (int pl = 0; pl < 24; ++pl) o[(size_t)pl * RR + pos] = pl == plane ? 1.0f : 0.0f;
    }
  } else {
    const int P = R + 2;
    uint16_t *o = out_nhwc + (size_t)g * P * P * 32;
    for (int pos = lane; pos < RR; pos += 64) {
      int plane = -1;
      const int i = pos / R, j = pos % R;
      if (live) {
        const uint8_t p = b->sq[rot90_src(R, k, i, j)];
        if (present(p)) plane = piece_plane(p, turn, c.rules);
      }
      uint32_t *row = reinterpret_cast<uint32_t *>(o + ((size_t)(i + 1) * P + (j + 1)) * 32);
      for (int w = 0; w < 16; ++w) {
        uint32_t v = 0;
        if (plane == 2 * w) v = one16; else if (plane == 2 * w + 1) v = (uint32_t)one16 << 16;
        row[w] = v;
      }
    }
  }
}

// legal-move mask (four_player_chess_board.py:36-56): dense 0/1 [n,A_ch,R,R] f32 in absolute coords
__global__ void __launch_bounds__(64) k_mask_from_moves(DevCfg c, const fpc_move *moves, const int *counts, int n,
                                                         float *out) {
  const int g = blockIdx.x;
  if (g >= n) return;
  const int lane = lane_id();
  float *o = out + (size_t)g * c.A;
  for (int i = lane; i < c.A; i += 64) o[i] = 0.f;
  __syncthreads();
  for (int k = lane; k < counts[g]; k += 64) o[moves[(size_t)g * FPC_MAX_MOVES + k].flat] = 1.0f;
}

// ================================================================================================
// k_search_init: fresh root per game (mcts.py:29-32: Node(C, game, visit_count=1))
// ================================================================================================
__global__ void __launch_bounds__(64) k_search_init(Tree t, int G, const fpc_board *roots) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int lane = lane_id();
  {  // root state -> board-pool slot 0 of this game
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&roots[g]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&t.boards[(size_t)g * t.board_cap]);
    dst[lane] = src[lane];
    if (lane < 8) dst[64 + lane] = src[64 + lane];
  }
  if (lane != 0) return;
  const size_t nb = (size_t)g * t.node_cap;
  t.N[nb] = 1; t.W[nb] = 0.0; t.P[nb] = 0.f; t.mv[nb] = 0xffff; t.parent[nb] = -1; t.child0[nb] = -1; t.nch[nb] = 0;
  t.bslot[nb] = 0;
  t.nnodes[g] = 1; t.nboards[g] = 1; t.alive[g] = 1; t.sims_done[g] = 0; t.err[g] = 0;
  t.leaf_node[g] = -1; t.leaf_slot[g] = -1; t.leaf_turn[g] = 0; t.nlegal[g] = 0;
}

// root read-back (alphazero.py:104-110 reads GetChildren / GetFlatIndex / GetVisitCount)
__global__ void __launch_bounds__(64) k_root_children(Tree t, int G, int maxc, int *flat, int *visits, float *prior,
                                                      double *wsum, int *meta, fpc_board *roots_out) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int lane = lane_id();
  const size_t nb = (size_t)g * t.node_cap;
  const int c0 = t.child0[nb], nc = c0 < 0 ? 0 : (int)t.nch[nb];
  for (int k = lane; k < nc && k < maxc; k += 64) {
    const size_t o = (size_t)g * maxc + k;
    flat[o] = t.mv[nb + c0 + k]; visits[o] = t.N[nb + c0 + k]; prior[o] = t.P[nb + c0 + k]; wsum[o] = t.W[nb + c0 + k];
  }
  {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&t.boards[(size_t)g * t.board_cap]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&roots_out[g]);
    dst[lane] = src[lane];
    if (lane < 8) dst[64 + lane] = src[64 + lane];
  }
  if (lane == 0) { meta[g * 3 + 0] = t.N[nb]; meta[g * 3 + 1] = nc; meta[g * 3 + 2] = t.sims_done[g]; }
}

// training tuples (alphazero.py:104-112): root mailbox + side to move + sparse pi of a finished search
__global__ void __launch_bounds__(64) k_collect_tuples(Tree t, int G, const int *game_id, int ply, fpc_tuple *out) {
  const int g = blockIdx.x;
  if (g >= G) return;
  const int lane = lane_id();
  fpc_tuple *rec = out + g;
  const fpc_board *root = &t.boards[(size_t)g * t.board_cap];
  const size_t nb = (size_t)g * t.node_cap;
  const int c0 = t.child0[nb], nc = c0 < 0 ? 0 : (int)t.nch[nb];
  for (int i = lane; i < FPC_MAX_SQ; i += 64) rec->sq[i] = root->sq[i];
  for (int k = lane; k < FPC_TUPLE_MAXC; k += 64) {
    const bool in = k < nc;
    rec->flat[k] = in ? t.mv[nb + c0 + k] : (uint16_t)0;
    rec->visits[k] = in ? (uint16_t)t.N[nb + c0 + k] : (uint16_t)0;
  }
  if (lane < 44) rec->pad1[lane] = 0;
  if (lane == 0) {
    rec->turn = root->turn; rec->pad0 = 0;
    rec->n = (uint16_t)(nc < FPC_TUPLE_MAXC ? nc : FPC_TUPLE_MAXC);
    rec->z = 0.f; rec->game = game_id ? game_id[g] : g; rec->ply = ply;
  }
}

// z by (game, team of the side to move): alphazero.py:128-137 / :161-175
__global__ void __launch_bounds__(256) k_tuples_set_z(fpc_tuple *recs, int count, const int *game_id, const float *z0, const float *z1, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int game = recs[i].game;
  for (int k = 0; k < n; ++k)
    if (game_id[k] == game) { recs[i].z = (recs[i].turn & 1) ? z1[k] : z0[k]; return; }
}

// ================================================================================================
// k_select: get_expandable_leaves / Node::ChooseLeaf (mcts.py:18-26, node.cpp:19-78)
// ================================================================================================
// Node::Backpropagate (node.cpp:133-142): value_sum += (double)v ; visit_count += 1 ; v = -v (f32)
__device__ __forceinline__ void backprop_lane0(const Tree &t, size_t nb, int n, float v) {
  while (n >= 0) {
    t.W[nb + n] += (double)v;
    t.N[nb + n] += 1;
    v = -v;
    n = t.parent[nb + n];
  }
}

// BackpropagateNodes (node.cpp:118-126) along the descent path k_select recorded: path[len-1] is the
// leaf (+v), its parent gets -v, ... -- every node of the path is touched once, so the lanes of one
// wave update them independently instead of one lane chasing parent pointers.
__device__ __forceinline__ void backprop_path(const Tree &t, size_t nb, int g, float v, int len, int node_of_lane) {
  // len = path_len[g] and node_of_lane = path[lane] (lane < len, first 64 entries) were fetched by the caller together
  // with everything else whose address needs no other load
  const int *path = t.path + (size_t)g * t.path_cap;
  for (int k = lane_id(); k < len; k += 64) {
    const int node = k < 64 ? node_of_lane : path[k];
    const float sv = ((len - 1 - k) & 1) ? -v : v;
    t.W[nb + node] += (double)sv;
    t.N[nb + node] += 1;
  }
}

// One game's selection step, executed by one wave (all 64 lanes call; `s` is the wave's LDS image).
// The chosen leaf goes to leaf_node/leaf_slot/leaf_turn, or to their _nx twins when `to_next`.
__device__ inline void select_game(WaveLds &s, const DevCfg &c, const Tree &t, int g, double Cpuct, const double *logtab, bool to_next) {
  const int lane = lane_id();
  int *const leaf_node = to_next ? t.leaf_node_nx : t.leaf_node;
  int *const leaf_slot = to_next ? t.leaf_slot_nx : t.leaf_slot;
  int *const leaf_turn = to_next ? t.leaf_turn_nx : t.leaf_turn;
  const size_t nb = (size_t)g * t.node_cap;
  // Everything whose address depends on no other load goes out FIRST, in one round trip (a lone wave per CU pays the
  // full memory latency for every dependent load: the chain of round trips, not the arithmetic, is this function's
  // floor): the liveness flag, the root's fields, the board-pool fill level the leaf materialisation will need.
  const int is_alive = t.alive[g];
  int c0 = t.child0[nb], nc = t.nch[nb], Nn = t.N[nb];
  int slot = t.bslot[nb], parent_slot = -1;        // board-pool slot of node n / of its parent
  const int nboards0 = t.nboards[g];
  if (!is_alive) {                          // root already removed from the search (Q5)
    if (lane == 0) { leaf_node[g] = -1; leaf_slot[g] = -1; }
    return;
  }
  // ---- descent: SelectChild (node.cpp:49-78)
  //   ucb_i = W_i/N_i + C * sqrt( log(sqrt(N_parent)) / (1 + N_i) ) * P_i      (fp64, no contraction)
  //   strict '>' from -inf => lowest index wins ties, NaN never wins
  // Every level costs ONE dependent global round trip: the lanes that fetch the children's N/W/P also
  // fetch each child's (first child, child count, board-pool slot, move), so the chosen child's own children can
  // be requested as soon as the argmax is known, and at the leaf its slot, its parent's and its move are at hand.
  int n = 0, depth = 0, leaf_mv = 0xffff;
  bool fail = false;
  int *path = t.path + (size_t)g * t.path_cap;
  for (;;) {
    if (lane == 0 && depth < t.path_cap) path[depth] = n;
    ++depth;
    if (c0 < 0) break;
    const double L = logtab[Nn];
    const double sqrtNp = sqrt((double)Nn);
    double best = 0.0;
    int besti = -1, best_c0 = -1, best_nc = 0, best_N = 0, best_slot = -1, best_mv = 0xffff;
    for (int base = 0; base < nc; base += 64) {
      const int i = base + lane;
      double u = 0.0;
      bool valid = false;
      int Nc = 0, cc0 = -1, cnc = 0, cslot = -1, cmv = 0xffff;
      if (i < nc) {
        Nc = t.N[nb + c0 + i];
        double Wc = t.W[nb + c0 + i];
        double Pc = (double)t.P[nb + c0 + i];
#ifdef FPC_TREE_LDS_STAGE
        // A/B arm only (north_star: "per-game node N/W/P arrays staged in LDS"): the level's children pass through LDS
        // before the PUCT arithmetic reads them -- one more dependent round trip per level, no reuse to pay for it
        // (a level is read exactly once per simulation).  Measured in profiles/r05/ab_summary.md; the product keeps them in registers.
        {
          int *sN = reinterpret_cast<int *>(&s.scr[0][0][0]);
          double *sW = reinterpret_cast<double *>(&s.scr[0][0][0]) + 64;
          float *sP = reinterpret_cast<float *>(&s.scr[0][0][0]) + 64 + 128 + 64;
          sN[lane] = Nc; sW[lane] = Wc; sP[lane] = (float)Pc;
          __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the stores are in LDS
          Nc = *const_cast<volatile int *>(&sN[lane]); Wc = *const_cast<volatile double *>(&sW[lane]); Pc = (double)*const_cast<volatile float *>(&sP[lane]);
        }
#endif
        cc0 = t.child0[nb + c0 + i];
        cnc = t.nch[nb + c0 + i];
        cslot = t.bslot[nb + c0 + i];
        cmv = t.mv[nb + c0 + i];
        if (c.rules & FPC_RULES_PUCT) {
          // AlphaZero PUCT, the child's value seen from the parent: -W/N + C P sqrt(N_parent) / (1 + N)
          const double q = Nc > 0 ? -(Wc / (double)Nc) : 0.0;
          u = q + Cpuct * Pc * sqrtNp / (double)(1 + Nc);
        } else {
          const double q = Nc > 0 ? Wc / (double)Nc : 0.0;
          const double e = Cpuct * sqrt(L / (double)(1 + Nc));
          u = q + e * Pc;
        }
        valid = u > -__builtin_inf();
      }
      // argmax with the reference's rule (strict '>': the FIRST maximum wins): start at the lowest valid lane and jump
      // to the lowest lane that beats the current one until none does -- about ln(children) rounds of two scalar lane
      // reads, a compare and a ballot, instead of six butterfly rounds of three LDS-crossbar shuffles each
      const unsigned long long live = __ballot(valid);
      int idx = 0x7fffffff, cur = 0;
      double ubest = -__builtin_inf();
      if (live) {                                  // wave-uniform
        cur = (int)__ffsll((long long)live) - 1;
        for (;;) {
          ubest = wave_read(u, cur);
          const unsigned long long gt = __ballot(valid && u > ubest);
          if (!gt) break;
          cur = (int)__ffsll((long long)gt) - 1;
        }
        idx = base + cur;
      }
      const int w_c0 = wave_read(cc0, cur), w_nc = wave_read(cnc, cur), w_N = wave_read(Nc, cur), w_slot = wave_read(cslot, cur), w_mv = wave_read(cmv, cur);
      if (idx != 0x7fffffff && (besti < 0 || ubest > best)) { best = ubest; besti = idx; best_c0 = w_c0; best_nc = w_nc; best_N = w_N; best_slot = w_slot; best_mv = w_mv; }
    }
    if (besti < 0) { fail = true; break; }
    n = c0 + besti;
    c0 = best_c0; nc = best_nc; Nn = best_N;
    parent_slot = slot; slot = best_slot; leaf_mv = best_mv;
  }
  FPC_TS(5);
  if (fail) {                                // node.cpp:72-75 throws
    if (lane == 0) { t.err[g] |= ERR_SELECT; t.alive[g] = 0; leaf_node[g] = -1; leaf_slot[g] = -1; }
    return;
  }
  // ---- leaf state: the reference copies + MakeMoves a Board for every child at expansion time
  //      (node.cpp:90-91); here it is materialised the first time the node is reached.
  const fpc_board *pool = t.boards + (size_t)g * t.board_cap;
  if (slot < 0) {
    lds_load_board(&s, &pool[parent_slot]);
    int from;
    const int to = flat_to(c, leaf_mv, &from);
    const bool moved = make_move_wave(&s.b, from, to, c);
    if (lane == 0) {
      int e = moved ? 0 : ERR_MOVE;
      int ns = nboards0;                     // fetched with the root's fields; only this block ever changes it
      if (ns >= t.board_cap) { e |= ERR_CAP_BOARDS; ns = -1; } else { t.nboards[g] = ns + 1; t.bslot[nb + n] = ns; }
      s.first_legal = ns;                    // broadcast through LDS
      if (e) t.err[g] |= e;
    }
    __syncthreads();
    slot = s.first_legal;
    __syncthreads();
    if (slot < 0) {                          // board pool exhausted: the game leaves the search, nothing is overwritten
      if (lane == 0) { t.alive[g] = 0; leaf_node[g] = -1; leaf_slot[g] = -1; }
      return;
    }
  } else {
    lds_load_board(&s, &pool[slot]);
  }
  FPC_TS(6);
  // ---- GetGameResult (node.cpp:28-29) then, if in progress, GetLegalMoves
  //      (four_player_chess_board.py:38) on the same state, with their list reorderings
  wave_position_ops(&s, c, true, true, -1);
  FPC_TS(12);
  lds_store_board(&s, &t.boards[(size_t)g * t.board_cap + slot]);
  const int res = s.result;
  if (lane == 0 && s.errbits) t.err[g] |= s.errbits;
  if (res != FPC_IN_PROGRESS) {              // node.cpp:31-42: back up 0 / -1, drop the root (Q5)
    if (lane == 0) {
      backprop_lane0(t, nb, n, res == FPC_STALEMATE ? 0.0f : -1.0f);
      t.sims_done[g] += 1;
      t.alive[g] = 0;
      leaf_node[g] = -1;
      leaf_slot[g] = -1;
    }
    return;
  }
  const int nl = s.nlegal;
  uint16_t *lg = t.legal + (size_t)g * FPC_MAX_MOVES;
  for (int k = lane; k < nl; k += 64) lg[k] = s.lsorted[k];
  if (lane == 0) { leaf_node[g] = n; leaf_slot[g] = slot; leaf_turn[g] = s.b.turn; t.nlegal[g] = nl; t.path_len[g] = depth; }
  FPC_TS(13);
}

__global__ void __launch_bounds__(64) k_select(DevCfg c, Tree t, int G, double Cpuct, const double *logtab) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= G) return;
  select_game(s, c, t, g, Cpuct, logtab, false);
}

// ================================================================================================
// deterministic f32 exp shared (as a numeric SPEC, DESIGN.md "fpc_expf") with the test oracle:
// k = rint(x*log2e); r = x - k*ln2 (two fma steps); degree-7 Horner in fma; scale by 2^k.
// ================================================================================================
__device__ __forceinline__ float fpc_expf(float x) {
  if (x != x) return x;
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) return __builtin_inff();
  const float k = __builtin_rintf(x * 0x1.715476p+0f);
  float r = __builtin_fmaf(k, -0x1.62e4p-1f, x);
  r = __builtin_fmaf(k, -0x1.7f7d1cp-20f, r);
  float p = 0x1.a01a02p-13f;
  p = __builtin_fmaf(p, r, 0x1.6c16c2p-10f);
  p = __builtin_fmaf(p, r, 0x1.111112p-7f);
  p = __builtin_fmaf(p, r, 0x1.555556p-5f);
  p = __builtin_fmaf(p, r, 0x1.555556p-3f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.0f);
  p = __builtin_fmaf(p, r, 1.0f);
  const int ki = (int)k;
  const uint32_t bits = (uint32_t)(ki + 127) << 23;
  float sc;
  __builtin_memcpy(&sc, &bits, 4);
  return p * sc;
}

__device__ __forceinline__ float fdiv_rn(float a, float b) {
#ifdef FPC_EMUL
  return a / b;
#else
  return __fdiv_rn(a, b);
#endif
}

// Shared end of both expand kernels (one wave): s.pri[0..nl) = unnormalised legal probabilities in
// ascending flat order, s.lsorted the flat indices.  Legal mass (sequential ascending f32 sum),
// policy error, BackpropagateNodes (mcts.py:78) before ExpandNodes (mcts.py:79), children appended.
struct ExpandPre {     // what expand_game / expand_legal_game fetch up front for expand_finish (one round trip with their own loads)
  float v;             // value[g]
  int path_len, path_node, nnodes;
};
__device__ __forceinline__ ExpandPre expand_prefetch(const Tree &t, int g, const float *value) {
  ExpandPre e;
  const int lane = lane_id();
  e.v = value[g];
  e.path_len = t.path_len[g];
  e.path_node = lane < t.path_cap ? t.path[(size_t)g * t.path_cap + lane] : 0;
  e.nnodes = t.nnodes[g];
  return e;
}
__device__ __forceinline__ void expand_finish(WaveLds &s, const Tree &t, int g, size_t nb, int n, int nl, bool nan, const ExpandPre &pre) {
  const int lane = lane_id();
  if (lane == 0) {
    float T = 0.f;
    for (int j = 0; j < nl; ++j) T = T + s.pri[j];
    s.errbits = (nan || !(T > 0.f)) ? ERR_POLICY : 0;
    s.scal_f = T;
  }
  __syncthreads();
  if (s.errbits) {                          // the reference would expand all A indices and throw
    if (lane == 0) { t.err[g] |= ERR_POLICY; t.alive[g] = 0; }
    return;
  }
  const float T = s.scal_f;
  // root Dirichlet noise (N4): prior' = (1 - eps) prior + eps g_j / sum g over the root's legal moves,
  // g = caller-supplied Gamma(alpha) draws; sequential ascending f32 sum, IEEE division
  const bool noisy = n == 0 && t.noise != nullptr;
  if (noisy) {
    const float *gm = t.noise + (size_t)g * FPC_MAX_MOVES;
    __syncthreads();
    if (lane == 0) {
      float sg = 0.f;
      for (int j = 0; j < nl; ++j) sg = sg + gm[j];
      s.scal_f = sg;
    }
    __syncthreads();
  }
  const float SG = s.scal_f;
  const bool add_noise = noisy && SG > 0.f;      // rows without positive mass (never uploaded, all-zero draws) leave the priors alone
  // BackpropagateNodes (mcts.py:78) before ExpandNodes (mcts.py:79)
  backprop_path(t, nb, g, pre.v, pre.path_len, pre.path_node);
  if (lane == 0) t.sims_done[g] += 1;
  // children: ascending flat order, entries with prior == 0 dropped (torch.nonzero, mcts.py:84)
  const int base_node = pre.nnodes;
  int created = 0;
  for (int b0 = 0; b0 < nl; b0 += 64) {
    const int j = b0 + lane;
    float pr = 0.f;
    bool nz = false;
    if (j < nl) {
      pr = fdiv_rn(s.pri[j], T);
      if (add_noise) pr = (1.0f - t.noise_eps) * pr + t.noise_eps * fdiv_rn(t.noise[(size_t)g * FPC_MAX_MOVES + j], SG);
      nz = pr != 0.f;
    }
    const unsigned long long bal = __ballot(nz);
    if (nz) {
      const int k = base_node + created + __popcll(bal & ((1ull << lane) - 1ull));
      if (k < t.node_cap) {
        t.N[nb + k] = 1; t.W[nb + k] = 0.0; t.P[nb + k] = pr; t.mv[nb + k] = s.lsorted[j];
        t.parent[nb + k] = n; t.child0[nb + k] = -1; t.nch[nb + k] = 0; t.bslot[nb + k] = -1;
      }
    }
    created += __popcll(bal);
  }
  if (lane == 0) {
    if (base_node + created > t.node_cap) { t.err[g] |= ERR_CAP_NODES; t.alive[g] = 0; }
    else if (created > 0) {
      t.child0[nb + n] = base_node; t.nch[nb + n] = (uint16_t)created; t.nnodes[g] = base_node + created;
    }
  }
}

// ================================================================================================
// Softmax statistics of a logits row, by chunks of 1024 logits (= 256 threads x float4) -- the first half
// of the numeric spec of mcts.py:67.  Chunk c of a row: m_c = its maximum,
// s_c = sum of fpc_expf(x - m_c) in the order "thread t owns float4 group t of the chunk, adds x,y,z,w
// ascending; xor butterfly 32..1 inside each 64-lane wave; ((w0+w1)+w2)+w3", flag = a NaN was seen.
// The policy Linear's k_fc_reduce produces the records while it still holds the logits in registers
// (fpc_fc.h); k_softmax_partials produces the same records for logits handed in by an external evaluator.
// ================================================================================================
constexpr int SM_THREADS = 256;
constexpr int SM_REC = 4;              // floats per chunk record: max, sum, NaN flag, pad
constexpr int SM_MAXCH = 32;           // records per row (A <= 32768; 23 at 14x14)
// all SM_THREADS threads call; `red`: 8 floats of shared memory
__device__ __forceinline__ void softmax_chunk_stats(float4 v, bool valid, float *red, float *rec) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -__builtin_inff();
  bool nan = false;
  if (valid) {
    nan = (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    m = v.x > m ? v.x : m; m = v.y > m ? v.y : m; m = v.z > m ? v.z : m; m = v.w > m ? v.w : m;
  }
  for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(m, off); m = o > m ? o : m; }
  const bool wnan = __ballot(nan) != 0ull;
  if (lane == 0) { red[wave] = m; red[4 + wave] = wnan ? 1.f : 0.f; }
  __syncthreads();
  m = red[0];
  m = red[1] > m ? red[1] : m; m = red[2] > m ? red[2] : m; m = red[3] > m ? red[3] : m;
  const float nf = ((red[4] != 0.f) | (red[5] != 0.f) | (red[6] != 0.f) | (red[7] != 0.f)) ? 1.f : 0.f;
  __syncthreads();
  float part = 0.f;
  if (valid && m > -__builtin_inff()) {
    part = part + fpc_expf(v.x - m);
    part = part + fpc_expf(v.y - m);
    part = part + fpc_expf(v.z - m);
    part = part + fpc_expf(v.w - m);
  }
  for (int off = 32; off >= 1; off >>= 1) part = part + __shfl_xor(part, off);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) { rec[0] = m; rec[1] = ((red[0] + red[1]) + red[2]) + red[3]; rec[2] = nf; rec[3] = 0.f; }
}

// grid = G * nchunks blocks: block b -> (row b / nchunks, chunk b % nchunks)
__global__ void __launch_bounds__(SM_THREADS) k_softmax_partials(const float *logits, int A, int G, int nchunks, float *stats) {
  __shared__ float red[8];
  const int g = (int)blockIdx.x / nchunks, ch = (int)blockIdx.x % nchunks;
  if (g >= G) return;
  const int q = ch * SM_THREADS + (int)threadIdx.x;
  const bool valid = q * 4 < A;
  const float4 v = valid ? *reinterpret_cast<const float4 *>(logits + (size_t)g * A + 4 * q) : float4{0.f, 0.f, 0.f, 0.f};
  softmax_chunk_stats(v, valid, red, stats + ((size_t)g * SM_MAXCH + ch) * SM_REC);
}

// ================================================================================================
// k_expand: mcts.py:67-89 for one leaf per wave.
//   p = softmax(logits) over all A entries;  policy_abs[pl][r][c] = p[pl][rot90 by -turn0]  (Q6);
//   prior_j = p_src(j) / sum_legal p  for the ascending legal list; exact zeros get no child;
//   BackpropagateNodes(value) first (Q4), then children appended with N=1 (Q1), W=0.
// Numeric spec, second half: m = max_c m_c; S = sequential ascending sum over the
// chunks of s_c * fpc_expf(m_c - m) (a chunk of -inf only contributes 0); p_i = fpc_expf(l_i - m) * (1/S);
// the legal mass is a sequential ascending f32 sum.  The row itself is only touched at the legal moves.
// ================================================================================================
constexpr int EXPAND_THREADS = 64;
// Where a leaf's logits come from.  `dense` != null: a [G][A] f32 matrix (external evaluators; fpc_nn_forward).
// Else: the policy Linear's split-K partial sums as k_fc16 / k_fcw left them -- `part` [slab][Mtot][gw] f32 and the bias --
// which the fused search never combines into a dense matrix: the softmax records come from k_fc_reduce, and the
// expansion adds up the slabs of a column itself at the leaf's ~40 legal moves, in k_fc_reduce's order
// (bias, then the group's slabs ascending: the same f32 additions, the same bits).
constexpr int LOGIT_MAX_SLABS = 8;
struct LogitSrc {
  const float *dense;
  const float *part, *bias;
  int G1, s1, s2, Mtot;      // column groups [0, G1) own s1 slabs each, the rest s2 (plan_fc)
  int gw;                    // columns per group: 256 (k_fc / k_fc16) or 384 (k_fcw)
};
__device__ __forceinline__ float logit_at(const LogitSrc &L, int g, int A, int idx) {
  if (L.dense) return L.dense[(size_t)g * A + idx];
  const int j = L.gw == 256 ? idx >> 8 : idx / L.gw, col = idx - j * L.gw;   // column group (fpc_fc.h)
  const int base = j < L.G1 ? j * L.s1 : L.G1 * L.s1 + (j - L.G1) * L.s2, cnt = j < L.G1 ? L.s1 : L.s2;
  // all of a column's slabs are requested before the first is added (one round trip, not `cnt`): a group has at most
  // LOGIT_MAX_SLABS = 2 * FC_SPLITK of them (plan_fc)
  float p[LOGIT_MAX_SLABS];
#pragma unroll
  for (int k = 0; k < LOGIT_MAX_SLABS; ++k) p[k] = k < cnt ? L.part[((size_t)(base + k) * L.Mtot + g) * L.gw + col] : 0.f;
  float v = L.bias[idx];
#pragma unroll
  for (int k = 0; k < LOGIT_MAX_SLABS; ++k) v = k < cnt ? v + p[k] : v;
  return v;
}
__device__ inline void expand_game(WaveLds &s, const DevCfg &c, const Tree &t, int G, int g, const LogitSrc &logits, const float *stats,
                                   const float *value) {
  const int lane = lane_id();
  const size_t nb = (size_t)g * t.node_cap;
  const int nchunks = (c.A / 4 + SM_THREADS - 1) / SM_THREADS;     // A = (8R+8)*R*R is a multiple of 4 for even R
  const float *st = stats + (size_t)g * SM_MAXCH * SM_REC;
  const uint16_t *legal = t.legal + (size_t)g * FPC_MAX_MOVES;
  // ONE round trip for everything whose address depends on no other load (a lone wave per CU pays the full memory
  // latency per dependent load): the leaf, its legal list (first 64 entries), the chunk records, the leaf turns of the
  // first 64 games (the batch's first live leaf is almost always among them), value / path / node count.
  const int n = t.leaf_node[g];
  const int nl = t.nlegal[g];
  const int own_turn = t.leaf_turn[g];
  const int scan_node = lane < G ? t.leaf_node[lane] : -1;
  const int scan_turn = lane < G ? t.leaf_turn[lane] : 0;
  const int fl0 = legal[lane];                                    // lane < 64 <= FPC_MAX_MOVES: always in bounds
  float mc = -__builtin_inff(), sc = 0.f;
  bool nan = false;
  if (lane < nchunks) { mc = st[lane * SM_REC]; sc = st[lane * SM_REC + 1]; nan = st[lane * SM_REC + 2] != 0.f; }
  const ExpandPre pre = expand_prefetch(t, g, value);
  int turn0;
  if (c.rules & FPC_RULES_ROTATION) turn0 = own_turn;
  else {                                                          // the batch's FIRST live leaf (Q6)
    const unsigned long long bal = __ballot(scan_node >= 0);
    turn0 = bal ? wave_read(scan_turn, (int)__ffsll((long long)bal) - 1) : (G > 64 ? first_leaf_turn(t.leaf_node, t.leaf_turn, G) : 0);
  }
  if (n < 0) return;
  FPC_TS(1);
  float m = mc;
  for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(m, off); m = o > m ? o : m; }
  nan = __ballot(nan) != 0ull;
  const float term = (lane < nchunks && mc > -__builtin_inff()) ? sc * fpc_expf(mc - m) : 0.f;
  float S = 0.f;
  for (int k = 0; k < nchunks; ++k) S = S + wave_read(term, k);      // sequential ascending sum (numeric spec); k is wave-uniform
  const float inv = fdiv_rn(1.0f, S);
  FPC_TS(2);
  for (int j = lane; j < nl; j += 64) {
    const int fl = j < 64 ? fl0 : (int)legal[j];
    const int plane = fl / c.RR, pos = fl % c.RR;
    const int src = plane * c.RR + rot90_src(c.R, -turn0, pos / c.R, pos % c.R);
    s.pri[j] = fpc_expf(logit_at(logits, g, c.A, src) - m) * inv;
    s.lsorted[j] = (uint16_t)fl;
  }
  __syncthreads();
  FPC_TS(3);
  expand_finish(s, t, g, nb, n, nl, nan, pre);
}

__global__ void __launch_bounds__(EXPAND_THREADS) k_expand(DevCfg c, Tree t, int G, LogitSrc logits, const float *stats, const float *value) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= G) return;
  expand_game(s, c, t, G, g, logits, stats, value);
}

// k_expand of simulation step s followed, for the same game, by k_select of step s+1 (one launch and one
// dependent-kernel gap less per step).  Other blocks may still be expanding step s when this block's
// selection publishes its leaf, and strict mode reads the turn of the batch's FIRST live leaf across games
// (Q6), so the selection writes the _nx leaf arrays; the host swaps them in after the launch.
__global__ void __launch_bounds__(EXPAND_THREADS) k_expand_select(DevCfg c, Tree t, int G, LogitSrc logits, const float *stats,
                                                                  const float *value, double Cpuct, const double *logtab) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= G) return;
  FPC_TS(0);
  expand_game(s, c, t, G, g, logits, stats, value);
  __syncthreads();                           // the new children (global stores of other lanes) are visible to the descent
  FPC_TS(4);
  select_game(s, c, t, g, Cpuct, logtab, true);
}


// ================================================================================================
// k_expand_legal: the same step for the LEGAL-ONLY policy head (opt-in, fpc_set_policy_mode): the
// network evaluated the policy Linear only at the leaf's legal moves (k_policy_gemv, fpc_nn.h), so
// ll[g][j] is the logit of legal move j (ascending flat order, already taken through the inverse
// rotation).  prior_j = exp(l_j - max_legal) / sum_legal exp(l_k - max_legal): the full softmax's
// denominator cancels in mask-multiply + renormalise (mcts.py:67-76), so this equals the reference's
// priors up to f32 rounding -- except where a legal logit lies more than ~87 below the GLOBAL
// maximum, which the full softmax would flush to zero (child dropped / policy error) and this form
// cannot see.  One wave per game.
// ================================================================================================
__device__ inline void expand_legal_game(WaveLds &s, const DevCfg &c, const Tree &t, int g, const float *ll, const float *value) {
  const int lane = lane_id();
  const int n = t.leaf_node[g];
  const ExpandPre pre = expand_prefetch(t, g, value);
  if (n < 0) return;
  const size_t nb = (size_t)g * t.node_cap;
  const int nl = t.nlegal[g];
  const uint16_t *legal = t.legal + (size_t)g * FPC_MAX_MOVES;
  const float *lg = ll + (size_t)g * FPC_MAX_MOVES;
  float m = -__builtin_inff();
  bool nan = false;
  for (int j = lane; j < nl; j += 64) { const float v = lg[j]; nan |= v != v; m = v > m ? v : m; }
  for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(m, off); m = o > m ? o : m; }
  nan = __ballot(nan) != 0ull;
  for (int j = lane; j < nl; j += 64) {
    s.pri[j] = m == -__builtin_inff() ? 0.f : fpc_expf(lg[j] - m);     // every legal logit -inf: mass 0 -> policy error
    s.lsorted[j] = legal[j];
  }
  __syncthreads();
  expand_finish(s, t, g, nb, n, nl, nan, pre);
}

__global__ void __launch_bounds__(64) k_expand_legal(DevCfg c, Tree t, int G, const float *ll, const float *value) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= G) return;
  expand_legal_game(s, c, t, g, ll, value);
}

// the legal-only head's counterpart of k_expand_select
__global__ void __launch_bounds__(64) k_expand_legal_select(DevCfg c, Tree t, int G, const float *ll, const float *value, double Cpuct,
                                                            const double *logtab) {
  __shared__ WaveLds s;
  const int g = blockIdx.x;
  if (g >= G) return;
  expand_legal_game(s, c, t, g, ll, value);
  __syncthreads();
  select_game(s, c, t, g, Cpuct, logtab, true);
}

}  // namespace fpc
