/*
 * include/fpc_engine.h -- C-ABI of the MI355X-native batched self-play engine for 4-player chess.
 *
 * This is the drop-in boundary for the MCTS.search hot path of jorr3/Alphazero-4-player-chess.
 * The reference has no C-ABI of its own: its boundary is the pybind11 module `alphazero_cpp`
 * (/root/reference/src/cpp/wrapper.cpp:15-254) plus three Python files on top of it
 * (src/py/mcts.py, src/py/four_player_chess_board.py, src/py/fen_parser.py).  Every entry point
 * below names the reference interface it replaces.  The Python class surface the training loop
 * uses (Board, Move, Node, MCTS, FourPlayerChess ...) is rebuilt over these entry points by the
 * ctypes shim in alphazero-4-player-chess_amd/alphazero_cpp.py.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on
 * success or a negative fpc_status; fpc_last_error() gives the message (the shim raises
 * RuntimeError, mirroring the exception translator at wrapper.cpp:17-27).  All compute runs on the
 * GPU; there is no CPU fallback -- without a HIP device fpc_create fails with FPC_ENODEVICE.
 * One engine handle per GPU / per host thread (thread-compatible, not thread-safe).
 */
#ifndef FPC_ENGINE_H_
#define FPC_ENGINE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPC_MAX_SQ 196      /* 14 x 14 */
#define FPC_MAX_PL 16       /* pieces per colour kept in a piece list */
#define FPC_NO_SQ 255
#define FPC_MAX_MOVES 256   /* pseudo-legal moves per position handled on device (reference buffer: 300, engine/board.h:706) */

typedef enum fpc_status {
  FPC_OK = 0,
  FPC_EINVAL = -1,      /* bad argument */
  FPC_ENODEVICE = -2,   /* no HIP device / HIP runtime failure */
  FPC_ENOMEM = -3,
  FPC_ESELECT = -4,     /* node.cpp:72-75 "Failed to select a child." */
  FPC_EPOLICY = -5,     /* legal policy mass is 0/NaN: the reference expands every index and throws (mcts.py:76,84) */
  FPC_ECAPACITY = -6,   /* node/board pool or move buffer overflow (reference: abort(), engine/board.h:482-486) */
  FPC_EMOVE = -7,       /* engine/board.cpp:1046-1054 "piece missing for move" */
  FPC_EUNSUPPORTED = -8,/* reserved */
  FPC_ESTATE = -9,      /* call sequence error */
  FPC_EWEIGHTS = -10,   /* weight blob malformed / not loaded */
  FPC_ECOMM = -11       /* RCCL failure (library missing, communicator error): message names the ncclResult */
} fpc_status;

/* chess::GameResult, engine/board.h:438-444 */
enum { FPC_IN_PROGRESS = 0, FPC_WIN_RY = 1, FPC_WIN_BG = 2, FPC_STALEMATE = 3 };

/*
 * One position == chess::Board state (engine/board.h:696-706) as a 288-byte POD.
 * piece byte: 0 = empty, else 0x80 | colour<<5 | type<<2  (engine/board.h:101-104;
 * colour RED 0 BLUE 1 YELLOW 2 GREEN 3, type PAWN 0 .. KING 5).
 * pl[c][i] is the square (row*R+col) of the i-th entry of piece_list_[c] IN REFERENCE LIST ORDER:
 * that order is observable (it decides which move GetGameResult looks at first,
 * engine/board.cpp:904-925) and is mutated by MakeMove/UndoMove, so it is part of the state.
 */
typedef struct fpc_board {
  uint8_t sq[FPC_MAX_SQ];
  uint8_t pl[4][FPC_MAX_PL];
  uint8_t plen[4];
  uint8_t king[4];         /* king_locations_ (FPC_NO_SQ if captured) */
  uint8_t castle[4];       /* bit0 kingside, bit1 queenside (constant through tree/self-play moves) */
  uint8_t turn;
  uint8_t pad[15];
} fpc_board;               /* sizeof == 288 */

/* One legal move as the reference's Board::GetLegalMoves() reports it (board.cpp:94-118). */
typedef struct fpc_move {
  uint8_t from, to;
  uint8_t capture;         /* piece byte on `to` (0 none) */
  uint8_t promo;           /* 1 if the reference emits this move 4x (N,B,R,Q promotion variants, engine/board.cpp:82-88) */
  uint16_t flat;           /* Move::GetFlatIndex(), move.cpp:100-104 */
  uint16_t pad;
} fpc_move;

typedef struct fpc_config {
  int board_size;          /* rows_ == cols_  : 8 (literal snapshot) or 14      engine/board.h:22-23 */
  int invalid_area;        /* invalid_area    : 2 or 3                           engine/board.h:24    */
  int max_games;           /* concurrent games on this GPU (num_parallel_games, alphazero.py:303)    */
  int max_sims;            /* upper bound for num_searches (mcts.py:36)                              */
  int avg_children;        /* node pool = 1 + max_sims*avg_children per game; 0 -> default 96        */
  int device;              /* HIP device ordinal                                                     */
  int nn_dtype;            /* 0 = bf16, 1 = fp16 (MFMA operand type of the internal ResNet)          */
} fpc_config;

typedef struct fpc_engine fpc_engine;

int  fpc_create(const fpc_config *cfg, fpc_engine **out);
void fpc_destroy(fpc_engine *e);
const char *fpc_last_error(const fpc_engine *e);   /* e may be NULL: last create() failure */
int  fpc_abi_version(void);

/* ---- static geometry: fpchess::Board statics, board.cpp:9-14 / wrapper.cpp:176-181 ---- */
int fpc_num_action_channels(int board_size);       /* 4R+4C+8 */
int fpc_action_space_size(int board_size);
int fpc_is_legal_location(int board_size, int invalid_area, int row, int col);  /* engine/board.h:647-654 */
/* fpchess::Move codec, move.cpp:23-104.  fpc_flat_to_move returns FPC_NO_SQ in *to when the
 * target falls off the board (BoardLocation() "missing"). */
int fpc_move_flat_index(int board_size, int from, int to);   /* -1: GetIndex() throws */
int fpc_flat_to_move(int board_size, int flat, int *from, int *to);

/* ---- host-side construction: chess::Board::Board, engine/board.cpp:1172-1248 ----
 * (sq[i], piece[i]) in python-dict insertion order, exactly what pybind hands the reference ctor.
 * Reproduces the constructor's piece_list_ order (unordered_map iteration + std::sort). */
int fpc_board_from_dict(fpc_board *out, int board_size, int turn, const uint8_t *sq, const uint8_t *piece,
                        int n, const uint8_t *castle4 /* nullable */);

/* ---- batched position kernels (one wavefront per position) -------------------------------
 * boards[] live in HOST memory; they are uploaded, processed on the GPU and written back
 * (the reference mutates piece-list order in place in all of these). */
/* Board::GetLegalMoves  (board.cpp:94-118): moves[i*FPC_MAX_MOVES ..] in reference order */
int fpc_boards_legal_moves(fpc_engine *e, fpc_board *boards, int n, fpc_move *moves, int *counts);
/* Board::GetGameResult(opt_player) (engine/board.cpp:891-939); player[i] = -1 -> side to move */
int fpc_boards_game_result(fpc_engine *e, fpc_board *boards, int n, const int *player, int *results);
/* Board::TakeAction(Move(flat)) (board.cpp:234-239, move.cpp:39-61): out[i] = copy + MakeMove */
int fpc_boards_take_action(fpc_engine *e, const fpc_board *boards, const int *flat, int n, fpc_board *out);
/* Board::GetEncodedStates (board.cpp:305-356): out_host [n,24,R,R] f32, rotated by boards[0].turn */
int fpc_boards_encode(fpc_engine *e, const fpc_board *boards, int n, float *out_host);
/* FourPlayerChess.get_legal_moves_mask (four_player_chess_board.py:36-56): out_host [n,A_ch,R,R] f32 */
int fpc_boards_legal_mask(fpc_engine *e, fpc_board *boards, int n, float *out_host);
/* Board::GetAttackedSquaresPlayers / GetAttackedSquaresTeams / IsAttackedByPlayer (wrapper.cpp:204-206; board.cpp:120-232):
 * out_host [n][6][R*R] bytes, 1 = attacked -- maps 0..3 by colour (fpchess IsAttackedByPlayer: its own probe set, rays
 * through the cut corners), maps 4..5 by team (the engine's IsAttackedByTeam); every row x column of the array */
int fpc_boards_attack_maps(fpc_engine *e, const fpc_board *boards, int n, uint8_t *out_host);
/* Board::CalculateHeuristic (engine/board.cpp:1263-1292), pure host arithmetic on the POD */
int fpc_board_heuristic(const fpc_board *b, int team);

/* ---- MCTS.search (mcts.py:17-43) ---------------------------------------------------------
 * begin:   fresh root per game, visit_count = 1 (mcts.py:29-32); roots uploaded once.
 * Then either drive it step by step with an external evaluator (the reference's
 * `neural_net(x) -> (logits, value)` seam, mcts.py:65-66):
 *     for each of num_searches:  fpc_search_select -> evaluator -> fpc_search_expand
 * or let the engine run all simulations with its internal MFMA ResNet: fpc_search_run. */
int fpc_search_begin(fpc_engine *e, const fpc_board *roots, int n_games, double c_puct);
/* get_expandable_leaves + GetEncodedStates (mcts.py:18-26,65): selects one leaf per live game,
 * handles terminal leaves (node.cpp:31-42), encodes the leaves.  *n_live = leaves to evaluate.
 * enc_dev: DEVICE pointer, [n_games,24,R,R] f32 (slot g = game g; dead games are all-zero). */
int fpc_search_select(fpc_engine *e, int *n_live, const float **enc_dev);
/* softmax/ParseActionspace/mask/renormalise + BackpropagateNodes + ExpandNodes (mcts.py:67-89).
 * logits_dev [n_games, A] f32 and value_dev [n_games] f32 are DEVICE pointers (slot g = game g). */
int fpc_search_expand(fpc_engine *e, const float *logits_dev, const float *value_dev);
/* fpc_search_expand of this simulation followed by fpc_search_select of the next one, as ONE launch
 * (k_expand_select): what the loop mcts.py:36-38 does between two evaluator calls.  Same arguments and
 * results as the two calls it replaces; use plain fpc_search_expand after the last evaluation. */
int fpc_search_expand_select(fpc_engine *e, const float *logits_dev, const float *value_dev, int *n_live, const float **enc_dev);
/* all `sims` simulations with the internal network (weights from fpc_load_weights) */
int fpc_search_run(fpc_engine *e, int sims);
/* How fpc_search_run evaluates the policy head (net.py:22-26 + mcts.py:67-76):
 *   FPC_POLICY_FULL  (default) the whole Linear A -> A, softmax over all A outputs, mask, renormalise
 *                    -- the reference's arithmetic, op for op;
 *   FPC_POLICY_LEGAL the Linear only at the leaf's legal moves (the softmax denominator cancels in
 *                    mask + renormalise): the same priors up to f32 rounding, 1/50th of the weight
 *                    traffic.  Not bit-comparable with FULL; deviates where a legal logit lies ~87
 *                    below the global maximum (FULL flushes that child to zero).  Opt-in. */
enum { FPC_POLICY_FULL = 0, FPC_POLICY_LEGAL = 1 };
int fpc_set_policy_mode(fpc_engine *e, int mode);
/* ---- N4: root Dirichlet noise and the non-strict ("fixed") rule set ---------------
 * Strict reference semantics (every quirk of SURVEY 8a-Q, bit-exact with the reference) are the default.
 * fpc_set_rules switches individual corrections on for real training runs; none of them is covered by a
 * parity claim against the reference -- they are held against the oracle running the same rule set. */
enum {
  FPC_RULES_STRICT = 0,
  FPC_RULES_PUCT = 1,        /* Q2 + Q3: U = C P sqrt(N_parent) / (1 + N), child value seen from the parent (-W/N)   node.cpp:53-63 */
  FPC_RULES_ROTATION = 2,    /* Q6: every sample is rotated by its OWN side to move (encode and policy decode)       board.cpp:322,354-355 */
  FPC_RULES_PLANES = 4,      /* Q7: input plane 6*rel_colour + type, no -1 wrap                                       board.cpp:336 */
  FPC_RULES_FULL_MOVES = 8,  /* Q9: tree/self-play moves promote (to a queen), hop the rook when castling, update rights  node.cpp:87-92 */
  FPC_RULES_FIXED = 15
};
int fpc_set_rules(fpc_engine *e, int rules);
/* Root noise for the NEXT searches (mcts.py:45-56 defines add_dirichlet_noise and never calls it):
 * prior'_j = (1 - eps) prior_j + eps g_j / sum_k g_k over the root's legal moves j in ascending flat order,
 * gamma: host array [n_games][FPC_MAX_MOVES] of Gamma(alpha, 1) draws made (and seeded) by the caller;
 * NULL switches the noise off. */
int fpc_search_set_root_noise(fpc_engine *e, const float *gamma, int n_games, float eps);

/* Root read-back == what alphazero.py:104-110 reads through Node.GetChildren /
 * GetMoveMade().GetFlatIndex() / GetVisitCount().  Arrays are [n_games][max_children].
 * roots_out (nullable): the root states with the piece-list order the search left them in. */
int fpc_search_results(fpc_engine *e, fpc_board *roots_out, int *root_visits, int *n_children, int *sims_done,
                       int max_children, int *child_flat, int *child_visits, float *child_prior,
                       double *child_value_sum);
/* children of root child `child_idx` of game `game` (second tree level, for parity tests) */
int fpc_search_grandchildren(fpc_engine *e, int game, int child_idx, int max_children, int *n,
                             int *flat, int *visits);

/* ---- internal ResNet (net.py:6-63), BN folded, MFMA implicit-GEMM ------------------------ */
/* blob format: see alphazero-4-player-chess_amd/weights.py (header + folded tensors) */
int fpc_load_weights(fpc_engine *e, const void *blob, uint64_t nbytes);
/* forward only: enc_dev [n,24,R,R] f32 -> logits_dev [n,A] f32, value_dev [n] f32 (DEVICE pointers) */
int fpc_nn_forward(fpc_engine *e, const float *enc_dev, int n, float *logits_dev, float *value_dev);

/* ---- training tuples and their episode-end exchange ---------------------------------------
 * The reference keeps (state, pi, z) tuples as Python objects: Board::AppendToMemory(MemoryEntry(state,
 * action_probs)) per ply (alphazero.py:104-112), rewards assigned when the game ends
 * (handle_terminal_state, alphazero.py:53-78; heuristic scoring at max_game_length, :161-175).  Here a
 * tuple is a fixed-size POD built ON THE DEVICE from the finished search (root mailbox + side to move +
 * sparse pi = the root's (child flat index, visit count) pairs) and kept in a device buffer until the
 * episode ends.  Dense reference-shaped tensors (GetEncodedState [24,R,R], pi [A] = N / sum N) are
 * rebuilt from it on receipt (alphazero-4-player-chess_amd/tuples.py). */
#define FPC_TUPLE_MAXC 256
typedef struct fpc_tuple {
  uint8_t sq[FPC_MAX_SQ];        /* root mailbox (piece bytes as in fpc_board) */
  uint8_t turn;                  /* side to move at the root */
  uint8_t pad0;
  uint16_t n;                    /* number of (flat, visits) pairs */
  float z;                       /* filled by fpc_tuples_set_z */
  int32_t game, ply;             /* caller's game id and ply */
  uint16_t flat[FPC_TUPLE_MAXC];
  uint16_t visits[FPC_TUPLE_MAXC];
  uint8_t pad1[44];
} fpc_tuple;                     /* sizeof == 1280 */
int fpc_tuples_reserve(fpc_engine *e, int capacity);    /* device buffer for `capacity` tuples; drops collected ones */
int fpc_tuples_reset(fpc_engine *e);                    /* new episode: count = 0 */
/* after a finished search (fpc_search_run / the select-expand loop): one tuple per game of that search,
 * appended in game order.  game_id: host array [n_games] (NULL: 0..n_games-1). */
int fpc_collect_tuples(fpc_engine *e, const int *game_id, int ply);
/* z of every collected tuple whose game is game_id[i]: z_team[i][side-to-move team of the tuple]
 * (alphazero.py:128-137: +1 / -1 by team, quirk Q12; :161-175: +-heuristic) */
int fpc_tuples_set_z(fpc_engine *e, const int *game_id, const float *z_team0, const float *z_team1, int n);
int fpc_tuples_count(fpc_engine *e);
int fpc_tuples_read(fpc_engine *e, fpc_tuple *host_out, int first, int n);
/* RCCL, driven from the C++ host (one communicator per engine = per GPU; no torch involved):
 * rank 0 makes the 128-byte id and hands it to the other ranks by whatever channel the caller has. */
int fpc_comm_available(void);     /* 0 when librccl is bound in this process, else FPC_ECOMM (fpc_last_error(NULL) says why):
                                   * lets every rank agree on the exchange path BEFORE any of them enters ncclCommInitRank */
int fpc_comm_unique_id(void *id128);
int fpc_comm_init(fpc_engine *e, const void *id128, int rank, int world);
int fpc_comm_destroy(fpc_engine *e);
/* episode end: one ncclAllGather of the per-rank counts and one of the max-padded tuple
 * arrays over xGMI.  counts_out: host [world].  The gathered tuples stay on the device, rank-major
 * (rank r's tuples first .. counts_out[r] of them), and are read with fpc_gathered_read. */
int fpc_allgather_tuples(fpc_engine *e, int *counts_out, int *total_out);
/* Between the two sits an 8-byte status all-gather that ALWAYS runs: a rank on which anything local failed (an upload, a
 * read-back, growing a buffer) still goes through the counts and the status collective, every rank then leaves before
 * the payload collective -- the failing one with its own error, the others with FPC_ECOMM naming it -- and the
 * communicator stays usable.  Only an error of an RCCL call itself, or a failed read-back of the status words, ends
 * with the communicator aborted (ncclCommAbort) and fpc_comm_init needed again.
 * TEST HOOK: the next fpc_allgather_tuples treats one of its own HIP calls as failed -- 1 counts upload, 2 counts
 * read-back, 3 send-buffer growth, 4 status upload, 5 status read-back; 0 clears it. */
int fpc_debug_comm_fault(fpc_engine *e, int point);
int fpc_gathered_read(fpc_engine *e, fpc_tuple *host_out, int first, int n);

/* ---- measurement hooks (bench.py) -------------------------------------------------------- */
typedef struct fpc_stats {
  /* HIP-event time (events recorded on the engine's stream, resolved in fpc_search_results)
   * accumulated since fpc_stats_reset: select+encode | residual tower incl. head convs |
   * policy Linear (+ split-K reduce) | expand+backup.  Every 16th simulation step is timed and its
   * intervals count 16x (the events themselves cost time) */
  double ms_select, ms_tower, ms_fc, ms_expand;
  uint64_t launches_select, launches_nn, launches_expand;
  uint64_t sims;                        /* leaf evaluations + terminal backups */
  uint64_t nodes;                       /* nodes allocated */
} fpc_stats;
int fpc_stats_get(fpc_engine *e, fpc_stats *out);
int fpc_stats_reset(fpc_engine *e);
int fpc_set_timing(fpc_engine *e, int enabled);  /* HIP events around each stage (adds syncs) */
const char *fpc_nn_kernel(fpc_engine *e);        /* name of the kernel that runs the residual tower for the loaded weights:
                                                  * "k_towerw" (hidden 256; hidden 128 off the 14x14 board), "k_towerc" (hidden 128, 14x14: k_tower's
                                                  * skeleton on the compact image), "k_tower" (developer knobs), "k_conv3x3" (per layer), "" before fpc_load_weights */
void *fpc_stream(fpc_engine *e);                 /* hipStream_t the engine launches on */

#ifdef __cplusplus
}
#endif
#endif /* FPC_ENGINE_H_ */
