/*
 * oracle/fpc_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see fpc_oracle.h header).
 *
 * Scalar, single-threaded CPU restatement of the reference hot path.  Parity status: PINNED by
 * tests/test_oracle_golden.py against golden vectors generated from the real reference
 * (oracle/gen_golden.py, oracle/_ref).  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/cpp unless stated).
 */
#include "fpc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <unordered_map>
#include <vector>

namespace {

enum { PAWN = 0, KNIGHT = 1, BISHOP = 2, ROOK = 3, QUEEN = 4, KING = 5, NO_PIECE = 6 };
enum { RED = 0, BLUE = 1, YELLOW = 2, GREEN = 3 };
enum { RED_YELLOW = 0, BLUE_GREEN = 1 };
enum { IN_PROGRESS = 0, WIN_RY = 1, WIN_BG = 2, STALEMATE = 3 };

inline bool present(uint8_t p) { return (p & 0x80) != 0; }
inline int colour_of(uint8_t p) { return (p >> 5) & 3; }
inline int type_of(uint8_t p) { return (p >> 2) & 7; }
inline uint8_t mk_piece(int colour, int type) { return (uint8_t)(0x80 | (colour << 5) | (type << 2)); }
inline int team_of_colour(int c) { return (c == RED || c == YELLOW) ? RED_YELLOW : BLUE_GREEN; }  // engine/board.h:64-67
inline int team_of(uint8_t p) { return team_of_colour(colour_of(p)); }
inline int other_team(int t) { return t == RED_YELLOW ? BLUE_GREEN : RED_YELLOW; }  // engine/board.cpp:1526
inline int next_player(int c) { return (c + 1) & 3; }   // engine/board.cpp:1299-1313
inline int prev_player(int c) { return (c + 3) & 3; }   // engine/board.cpp:1331-1345

// engine/board.h:647-654
inline bool legal_loc(int R, int INV, int row, int col) {
  int mx = R - 1;
  if (row < 0 || row > mx || col < 0 || col > mx) return false;
  if (row < INV && (col < INV || col > mx - INV)) return false;
  if (row > mx - INV && (col < INV || col > mx - INV)) return false;
  return true;
}
inline bool in_array(int R, int row, int col) { return row >= 0 && row < R && col >= 0 && col < R; }

// ---- piece list primitives: engine/board.cpp:977-1014 ----
void set_piece(orc_board *b, int sq, uint8_t piece) {
  b->sq[sq] = piece;
  int c = colour_of(piece);
  if (b->plen[c] < ORC_MAX_PL) b->pl[c][b->plen[c]++] = (uint8_t)sq;
  if (type_of(piece) == KING) b->king[c] = (uint8_t)sq;
}
void remove_piece(orc_board *b, int sq) {
  uint8_t piece = b->sq[sq];
  b->sq[sq] = 0;
  int c = colour_of(piece);
  for (int i = 0; i < b->plen[c]; ++i) {
    if (b->pl[c][i] == sq) {
      for (int j = i; j + 1 < b->plen[c]; ++j) b->pl[c][j] = b->pl[c][j + 1];
      b->plen[c]--;
      break;
    }
  }
  if (type_of(piece) == KING) b->king[c] = ORC_NO_SQ;
}

struct MoveBuf {
  orc_move *buf;
  int pos, cap;
  void add(const orc_move &m) {
    if (pos < cap) buf[pos] = m;  // reference aborts on overflow (engine/board.h:482-486)
    pos++;
  }
};

orc_move mk_move(int from, int to, uint8_t capture) {
  orc_move m;
  m.from = (uint8_t)from; m.to = (uint8_t)to; m.capture = capture; m.promo = NO_PIECE;
  m.rook_from = ORC_NO_SQ; m.rook_to = ORC_NO_SQ; m.init_rights = 0; m.new_rights = 0;
  return m;
}

// engine/board.cpp:606-777 with limit=1 (IsAttackedByTeam, :779-787)
bool attacked_by_team(const orc_board *b, int R, int INV, int team, int sq) {
  int loc_row = sq / R, loc_col = sq % R;
  // rooks & queens: rays bounded by the ARRAY, not the cut corners (:632, quirk Q14)
  for (int do_incr_row = 0; do_incr_row < 2; ++do_incr_row)
    for (int pos_incr = 0; pos_incr < 2; ++pos_incr) {
      int ri = do_incr_row ? (pos_incr ? 1 : -1) : 0;
      int ci = do_incr_row ? 0 : (pos_incr ? 1 : -1);
      int row = loc_row + ri, col = loc_col + ci;
      while (in_array(R, row, col)) {
        uint8_t p = b->sq[row * R + col];
        if (present(p)) {
          if (team_of(p) == team && (type_of(p) == ROOK || type_of(p) == QUEEN)) return true;
          break;
        }
        row += ri; col += ci;
      }
    }
  // bishops & queens: bounded by IsLegalLocation (:658)
  for (int pr = 0; pr < 2; ++pr)
    for (int pc = 0; pc < 2; ++pc) {
      int ri = pr ? 1 : -1, ci = pc ? 1 : -1;
      int row = loc_row + ri, col = loc_col + ci;
      while (legal_loc(R, INV, row, col)) {
        uint8_t p = b->sq[row * R + col];
        if (present(p)) {
          if (team_of(p) == team && (type_of(p) == BISHOP || type_of(p) == QUEEN)) return true;
          break;
        }
        row += ri; col += ci;
      }
    }
  // knights: all 8 offsets (:676-694)
  for (int row_less = 0; row_less < 2; ++row_less)
    for (int pr = 0; pr < 2; ++pr) {
      int row = loc_row + (row_less ? (pr ? 1 : -1) : (pr ? 2 : -2));
      for (int pc = 0; pc < 2; ++pc) {
        int col = loc_col + (row_less ? (pc ? 2 : -2) : (pc ? 1 : -1));
        if (legal_loc(R, INV, row, col)) {
          uint8_t p = b->sq[row * R + col];
          if (present(p) && team_of(p) == team && type_of(p) == KNIGHT) return true;
        }
      }
    }
  // pawns (:697-750): array bounds only
  for (int pr = 0; pr < 2; ++pr) {
    int row = pr ? loc_row + 1 : loc_row - 1;
    if (row < 0 || row >= R) continue;
    for (int pc = 0; pc < 2; ++pc) {
      int col = pc ? loc_col + 1 : loc_col - 1;
      if (col < 0 || col >= R) continue;
      uint8_t p = b->sq[row * R + col];
      if (present(p) && team_of(p) == team && type_of(p) == PAWN) {
        bool att = false;
        switch (colour_of(p)) {
          case RED: att = pr != 0; break;
          case BLUE: att = pc == 0; break;
          case YELLOW: att = pr == 0; break;
          case GREEN: att = pc != 0; break;
        }
        if (att) return true;
      }
    }
  }
  // kings (:753-772)
  for (int dr = -1; dr < 2; ++dr)
    for (int dc = -1; dc < 2; ++dc) {
      if (!dr && !dc) continue;
      int row = loc_row + dr, col = loc_col + dc;
      if (legal_loc(R, INV, row, col)) {
        uint8_t p = b->sq[row * R + col];
        if (present(p) && team_of(p) == team && type_of(p) == KING) return true;
      }
    }
  return false;
}

// engine/board.cpp:941-951
bool king_in_check(const orc_board *b, int R, int INV, int colour) {
  int k = b->king[colour];
  if (k == ORC_NO_SQ) return false;
  return attacked_by_team(b, R, INV, other_team(team_of_colour(colour)), k);
}

// engine/board.cpp:47-93
void add_pawn_moves(MoveBuf &mv, int R, int from, int to, int colour, uint8_t capture) {
  bool promo = false;
  int trow = to / R, tcol = to % R;
  switch (colour) {
    case RED: promo = trow == R / 4; break;
    case BLUE: promo = tcol == 3 * R / 4; break;
    case YELLOW: promo = trow == 3 * R / 4; break;
    case GREEN: promo = tcol == R / 4; break;
  }
  orc_move m = mk_move(from, to, capture);
  if (promo) {
    for (int t = KNIGHT; t <= QUEEN; ++t) { m.promo = (uint8_t)t; mv.add(m); }
  } else {
    mv.add(m);
  }
}

// engine/board.cpp:97-177
void pawn_moves(const orc_board *b, int R, int INV, MoveBuf &mv, int from, uint8_t piece) {
  int colour = colour_of(piece), team = team_of(piece);
  int frow = from / R, fcol = from % R;
  int dr = 0, dc = 0;
  bool not_moved = false;
  switch (colour) {
    case RED: dr = -1; not_moved = frow == R - 2; break;
    case BLUE: dc = 1; not_moved = fcol == 1; break;
    case YELLOW: dr = 1; not_moved = frow == 1; break;
    case GREEN: dc = -1; not_moved = fcol == R - 2; break;
  }
  int trow = frow + dr, tcol = fcol + dc;
  if (legal_loc(R, INV, trow, tcol)) {
    if (!present(b->sq[trow * R + tcol])) {
      add_pawn_moves(mv, R, from, trow * R + tcol, colour, 0);
      if (not_moved) {
        int r2 = frow + 2 * dr, c2 = fcol + 2 * dc;
        // reference reads GetPiece(to) without a legality check (:143-145, quirk Q15); an
        // out-of-array square cannot occur for R >= 4 from the not_moved line.
        if (in_array(R, r2, c2) && !present(b->sq[r2 * R + c2]))
          add_pawn_moves(mv, R, from, r2 * R + c2, colour, 0);
      }
    }
  }
  bool check_cols = team == RED_YELLOW;
  for (int incr = 0; incr < 2; ++incr) {
    int crow = frow + dr, ccol = fcol + dc;
    if (check_cols) ccol += incr == 0 ? -1 : 1; else crow += incr == 0 ? -1 : 1;
    if (legal_loc(R, INV, crow, ccol)) {
      uint8_t o = b->sq[crow * R + ccol];
      if (present(o) && team_of(o) != team) add_pawn_moves(mv, R, from, crow * R + ccol, colour, o);
    }
  }
}

// engine/board.cpp:179-207 (loop bound is invalid_area: quirk Q8)
void knight_moves(const orc_board *b, int R, int INV, MoveBuf &mv, int from, uint8_t piece) {
  int frow = from / R, fcol = from % R;
  for (int prs = 0; prs < 2; ++prs)
    for (int adr = 1; adr < INV; ++adr) {
      int dr = prs > 0 ? adr : -adr;
      for (int pcs = 0; pcs < 2; ++pcs) {
        int adc = adr == 1 ? 2 : 1;
        int dc = pcs > 0 ? adc : -adc;
        int row = frow + dr, col = fcol + dc;
        if (legal_loc(R, INV, row, col)) {
          uint8_t cap = b->sq[row * R + col];
          if (!present(cap) || team_of(cap) != team_of(piece)) mv.add(mk_move(from, row * R + col, cap));
        }
      }
    }
}

// engine/board.cpp:209-238
void incr_moves(const orc_board *b, int R, int INV, MoveBuf &mv, uint8_t piece, int from, int ir, int ic,
                uint8_t init_rights, uint8_t new_rights) {
  int row = from / R + ir, col = from % R + ic;
  while (legal_loc(R, INV, row, col)) {
    uint8_t cap = b->sq[row * R + col];
    orc_move m = mk_move(from, row * R + col, cap);
    m.init_rights = init_rights; m.new_rights = new_rights;
    if (!present(cap)) {
      mv.add(m);
    } else {
      if (team_of(cap) != team_of(piece)) mv.add(m);
      break;
    }
    row += ir; col += ic;
  }
}

// engine/board.cpp:240-254
void bishop_moves(const orc_board *b, int R, int INV, MoveBuf &mv, int from, uint8_t piece) {
  for (int pr = 0; pr < 2; ++pr)
    for (int pc = 0; pc < 2; ++pc) incr_moves(b, R, INV, mv, piece, from, pr ? 1 : -1, pc ? 1 : -1, 0, 0);
}

// engine/board.cpp:1474-1524 + :23-30
int rook_location_type(int R, int INV, int colour, int sq) {  // 0 kingside, 1 queenside, -1 none
  int ks, qs;
  switch (colour) {
    case RED: ks = (R - 1) * R + (R - 4); qs = (R - 1) * R + INV; break;
    case BLUE: ks = (R - 4) * R + 0; qs = INV * R + 0; break;
    case YELLOW: ks = 0 * R + INV; qs = 0 * R + (R - 4); break;
    default: ks = INV * R + (R - 1); qs = (R - 4) * R + (R - 1); break;
  }
  if (sq == ks) return 0;
  if (sq == qs) return 1;
  return -1;
}

// engine/board.cpp:256-302
void rook_moves(const orc_board *b, int R, int INV, MoveBuf &mv, int from, uint8_t piece) {
  uint8_t init = 0, nw = 0;
  int ct = rook_location_type(R, INV, colour_of(piece), from);
  if (ct >= 0) {
    uint8_t cur = b->castle[colour_of(piece)];
    bool K = cur & 1, Q = cur & 2;
    if (K || Q) {
      if (ct == 0) {
        if (K) { init = 0x80 | cur; nw = 0x80 | (Q ? 2 : 0); }
      } else {
        if (Q) { init = 0x80 | cur; nw = 0x80 | (K ? 1 : 0); }
      }
    }
  }
  for (int dpi = 0; dpi < 2; ++dpi) {
    int incr = dpi > 0 ? 1 : -1;
    for (int dir = 0; dir < 2; ++dir) {
      int ir = dir > 0 ? incr : 0, ic = dir > 0 ? 0 : incr;
      incr_moves(b, R, INV, mv, piece, from, ir, ic, init, nw);
    }
  }
}

// engine/board.cpp:313-466
void king_moves(const orc_board *b, int R, int INV, MoveBuf &mv, int from, uint8_t piece) {
  int colour = colour_of(piece);
  uint8_t cur = b->castle[colour];
  uint8_t init = 0x80 | cur, nw = 0x80;  // CastlingRights(false,false) is Present
  int frow = from / R, fcol = from % R;
  for (int dr = -1; dr < 2; ++dr)
    for (int dc = -1; dc < 2; ++dc) {
      if (!dr && !dc) continue;
      int row = frow + dr, col = fcol + dc;
      if (legal_loc(R, INV, row, col)) {
        uint8_t cap = b->sq[row * R + col];
        if (!present(cap) || team_of(cap) != team_of(piece)) {
          orc_move m = mk_move(from, row * R + col, cap);
          m.init_rights = init; m.new_rights = nw;
          mv.add(m);
        }
      }
    }
  int oteam = other_team(team_of(piece));
  for (int is_k = 0; is_k < 2; ++is_k) {
    bool allowed = is_k ? (cur & 1) : (cur & 2);
    if (!allowed) continue;
    int ur = 0, uc = 0;  // unit step towards the rook
    switch (colour) {
      case RED: uc = is_k ? 1 : -1; break;
      case BLUE: ur = is_k ? 1 : -1; break;
      case YELLOW: uc = is_k ? -1 : 1; break;
      default: ur = is_k ? -1 : 1; break;
    }
    int nb = is_k ? 2 : 3;
    int between[3];
    bool ok = true;
    for (int i = 0; i < nb; ++i) {
      int row = frow + ur * (i + 1), col = fcol + uc * (i + 1);
      if (!in_array(R, row, col)) { ok = false; break; }
      between[i] = row * R + col;
    }
    int rrow = frow + ur * (nb + 1), rcol = fcol + uc * (nb + 1);
    if (!ok || !in_array(R, rrow, rcol)) continue;  // reference would read out of bounds here
    uint8_t rook = b->sq[rrow * R + rcol];
    if (!present(rook) || type_of(rook) != ROOK || team_of(rook) != team_of(piece)) continue;
    bool piece_between = false;
    for (int i = 0; i < nb; ++i) if (present(b->sq[between[i]])) { piece_between = true; break; }
    if (piece_between) continue;
    if (!attacked_by_team(b, R, INV, oteam, between[0]) && !attacked_by_team(b, R, INV, oteam, from)) {
      orc_move m = mk_move(from, between[1], 0);
      m.rook_from = (uint8_t)(rrow * R + rcol); m.rook_to = (uint8_t)between[0];
      m.init_rights = init; m.new_rights = nw;
      mv.add(m);
    }
  }
}

// engine/board.cpp:846-889
int pseudo_legal(const orc_board *b, int R, int INV, orc_move *out, int cap) {
  MoveBuf mv{out, 0, cap};
  int c = b->turn;
  if (b->king[c] == ORC_NO_SQ) return 0;
  for (int i = 0; i < b->plen[c]; ++i) {
    int sq = b->pl[c][i];
    uint8_t p = b->sq[sq];
    switch (type_of(p)) {
      case PAWN: pawn_moves(b, R, INV, mv, sq, p); break;
      case KNIGHT: knight_moves(b, R, INV, mv, sq, p); break;
      case BISHOP: bishop_moves(b, R, INV, mv, sq, p); break;
      case ROOK: rook_moves(b, R, INV, mv, sq, p); break;
      case QUEEN: bishop_moves(b, R, INV, mv, sq, p); rook_moves(b, R, INV, mv, sq, p); break;  // :304-311
      case KING: king_moves(b, R, INV, mv, sq, p); break;
    }
  }
  return mv.pos;
}

// engine/board.cpp:1028-1096
int make_move(orc_board *b, const orc_move *m) {
  uint8_t piece = b->sq[m->from];
  if (m->to >= ORC_MAX_SQ) return -1;  // Move(flat) pointing off the board: reference UB
  uint8_t cap = b->sq[m->to];
  if (present(cap)) remove_piece(b, m->to);
  if (!present(piece)) return -1;  // throws "piece missing" (:1046-1054) AFTER removing the capture
  remove_piece(b, m->from);
  if (m->promo != NO_PIECE) set_piece(b, m->to, mk_piece(b->turn, m->promo));
  else set_piece(b, m->to, piece);
  if (m->rook_from != ORC_NO_SQ && m->rook_to != ORC_NO_SQ) {
    uint8_t rook = b->sq[m->rook_from];
    remove_piece(b, m->rook_from);
    set_piece(b, m->rook_to, rook);
  }
  if (m->new_rights & 0x80) b->castle[b->turn] = m->new_rights & 3;
  b->turn = (uint8_t)next_player(b->turn);
  return 0;
}

// engine/board.cpp:1098-1160
void undo_move(orc_board *b, const orc_move *m) {
  int tb = prev_player(b->turn);
  uint8_t piece = b->sq[m->to];
  remove_piece(b, m->to);
  if (m->promo != NO_PIECE) set_piece(b, m->from, mk_piece(tb, PAWN));
  else set_piece(b, m->from, piece);
  if (present(m->capture)) set_piece(b, m->to, m->capture);
  if (m->rook_from != ORC_NO_SQ && m->rook_to != ORC_NO_SQ) {
    remove_piece(b, m->rook_to);
    set_piece(b, m->rook_from, mk_piece(tb, ROOK));
  }
  if (m->init_rights & 0x80) b->castle[tb] = m->init_rights & 3;
  b->turn = (uint8_t)tb;
}

// board.cpp:59-68 + :94-118
int legal_moves(orc_board *b, int R, int INV, orc_move *out, int cap) {
  orc_move buf[300];
  int n = pseudo_legal(b, R, INV, buf, 300);
  if (n > 300) n = 300;
  int k = 0;
  for (int i = 0; i < n; ++i) {
    int cur = b->turn;
    make_move(b, &buf[i]);
    bool safe = !king_in_check(b, R, INV, cur);
    undo_move(b, &buf[i]);
    if (safe) { if (k < cap) out[k] = buf[i]; k++; }
  }
  return k;
}

// engine/board.cpp:891-939, :962-975
int game_result(orc_board *b, int R, int INV, int player) {
  if (player < 0) player = b->turn;
  if (b->king[player] == ORC_NO_SQ) return team_of_colour(player) == RED_YELLOW ? WIN_BG : WIN_RY;
  orc_move buf[300];
  int n = pseudo_legal(b, R, INV, buf, 300);
  if (n > 300) n = 300;
  for (int i = 0; i < n; ++i) {
    make_move(b, &buf[i]);
    bool legal = !king_in_check(b, R, INV, player);
    int kc = IN_PROGRESS;
    if (present(buf[i].capture) && type_of(buf[i].capture) == KING)
      kc = team_of(buf[i].capture) == RED_YELLOW ? WIN_BG : WIN_RY;
    undo_move(b, &buf[i]);
    if (!legal) continue;
    if (kc != IN_PROGRESS) return kc;
    return IN_PROGRESS;
  }
  if (!king_in_check(b, R, INV, player)) return STALEMATE;
  return (player == RED || player == YELLOW) ? WIN_BG : WIN_RY;
}

// ---- move codec: move.cpp:13-104 ----
const int Q_OFF[8][2] = {{0, -1}, {-1, -1}, {-1, 0}, {-1, 1}, {0, 1}, {1, 1}, {1, 0}, {1, -1}};   // (dx=dcol, dy=drow)
const int N_OFF[8][2] = {{-2, -1}, {-2, 1}, {-1, -2}, {-1, 2}, {1, -2}, {1, 2}, {2, -1}, {2, 1}};

int move_plane(int R, int from, int to) {
  int dx = to % R - from % R, dy = to / R - from / R;
  int nq = R - 1;
  for (int i = 0; i < 8; ++i)
    for (int d = 1; d <= nq; ++d)
      if (Q_OFF[i][0] * d == dx && Q_OFF[i][1] * d == dy) return i * nq + (d - 1);
  for (int i = 0; i < 8; ++i)
    if (N_OFF[i][0] == dx && N_OFF[i][1] == dy) return 8 * nq + i;
  return -1;
}

// Move(action_plane, from) / Move(flat) : move.cpp:23-61; BoardLocation ctor engine/board.h:194-199
int plane_to(int R, int plane, int from) {
  int nq = R - 1, frow = from / R, fcol = from % R, trow, tcol;
  if (plane < 8 * nq) {
    int dir = plane / nq, dist = plane % nq;
    trow = frow + Q_OFF[dir][1] * (dist + 1);
    tcol = fcol + Q_OFF[dir][0] * (dist + 1);
  } else {
    int k = plane - 8 * nq;
    if (k >= 8) return ORC_NO_SQ;  // planes 8(R-1)+8.. are unaddressable (quirk Q11)
    trow = frow + N_OFF[k][1];
    tcol = fcol + N_OFF[k][0];
  }
  if (!in_array(R, trow, tcol)) return ORC_NO_SQ;
  return trow * R + tcol;
}

int rot90_src(int R, int k, int i, int j) {
  k = ((k % 4) + 4) % 4;
  switch (k) {
    case 0: return i * R + j;
    case 1: return j * R + (R - 1 - i);
    case 2: return (R - 1 - i) * R + (R - 1 - j);
    default: return (R - 1 - j) * R + i;
  }
}

}  // namespace

// ---- N4: the non-strict rule set and root noise, same definitions as include/fpc_engine.h FPC_RULES_* ----
namespace {
enum { RULES_PUCT = 1, RULES_ROTATION = 2, RULES_PLANES = 4, RULES_FULL_MOVES = 8 };
int g_rules = 0;
const float *g_noise = nullptr;   // [G][noise_stride] gamma draws per root child (ascending flat order)
int g_noise_stride = 0;
float g_noise_eps = 0.f;
}

extern "C" {

void orc_set_rules(int rules) { g_rules = rules; }
void orc_set_root_noise(const float *gamma, int stride, float eps) { g_noise = gamma; g_noise_stride = stride; g_noise_eps = eps; }

int orc_action_channels(int R) { return 4 * R + 4 * R + 8; }                 // board.cpp:11
int orc_action_size(int R) { return orc_action_channels(R) * R * R; }        // board.cpp:12
int orc_is_legal_location(int R, int INV, int row, int col) { return legal_loc(R, INV, row, col); }
int orc_move_plane(int R, int from, int to) { return move_plane(R, from, to); }
int orc_move_flat(int R, int from, int to) {                                 // move.cpp:100-104
  int p = move_plane(R, from, to);
  return p < 0 ? -1 : p * R * R + from;
}
int orc_flat_to_move(int R, int flat, int *from, int *to) {                  // move.cpp:39-61
  int plane = flat / (R * R), pos = flat % (R * R);
  *from = pos;
  *to = plane_to(R, plane, pos);
  return 0;
}

void orc_board_init(orc_board *b, int R, int turn) {
  (void)R;
  memset(b, 0, sizeof(*b));
  for (int c = 0; c < 4; ++c) b->king[c] = ORC_NO_SQ;
  b->turn = (uint8_t)turn;
}
int orc_board_add(orc_board *b, int R, int colour, int type, int sq) {
  (void)R;
  if (b->plen[colour] >= ORC_MAX_PL) return -1;
  set_piece(b, sq, mk_piece(colour, type));
  return 0;
}

// engine/board.cpp:1172-1248.  Iteration order of std::unordered_map is implementation-defined
// (quirk Q16); this restatement uses the same container, hash (engine/board.h:229-237), insertion
// sequence (pybind11 map_caster: reserve(n) then emplace in dict order) and std::sort comparator,
// so with the same libstdc++ it reproduces the reference order; pinned by golden GetPieces().
namespace {
struct LocHash { int R; size_t operator()(uint8_t s) const {
  size_t h = 14479 + 14593 * (size_t)(int8_t)(s / R); h += 24439 * (size_t)(int8_t)(s % R); return h; } };
}
void orc_board_from_dict(orc_board *b, int R, int turn, const uint8_t *sqs, const uint8_t *pieces, int n,
                         const uint8_t *castle4) {
  orc_board_init(b, R, turn);
  if (castle4) for (int c = 0; c < 4; ++c) b->castle[c] = castle4[c] & 3;
  std::unordered_map<uint8_t, uint8_t, LocHash> m(0, LocHash{R});
  m.reserve(n);
  for (int i = 0; i < n; ++i) m.emplace(sqs[i], pieces[i]);
  // pybind hands the map to fpchess::Board by value, which copies it (node order preserved),
  // then chess::Board iterates it (:1209).
  std::vector<std::pair<uint8_t, uint8_t>> lists[4];
  for (const auto &it : m) {
    b->sq[it.first] = it.second;
    lists[colour_of(it.second)].push_back({it.first, it.second});
    if (type_of(it.second) == KING) b->king[colour_of(it.second)] = it.first;
  }
  static const int score[6] = {1, 2, 3, 4, 5, 0};  // :1230-1236
  for (int c = 0; c < 4; ++c) {
    std::sort(lists[c].begin(), lists[c].end(), [](const std::pair<uint8_t, uint8_t> &a, const std::pair<uint8_t, uint8_t> &b2) {
      return score[type_of(a.second)] < score[type_of(b2.second)];
    });
    b->plen[c] = 0;
    for (auto &e : lists[c]) if (b->plen[c] < ORC_MAX_PL) b->pl[c][b->plen[c]++] = e.first;
  }
}

int orc_pseudo_legal(orc_board *b, int R, int INV, orc_move *out, int cap) { return pseudo_legal(b, R, INV, out, cap); }
int orc_legal_moves(orc_board *b, int R, int INV, orc_move *out, int cap) { return legal_moves(b, R, INV, out, cap); }
int orc_game_result(orc_board *b, int R, int INV, int player) { return game_result(b, R, INV, player); }
int orc_is_king_in_check(const orc_board *b, int R, int INV, int colour) { return king_in_check(b, R, INV, colour); }
int orc_is_attacked_by_team(const orc_board *b, int R, int INV, int team, int sq) { return attacked_by_team(b, R, INV, team, sq); }
// fpchess::Board::IsAttackedByPlayer (src/cpp/board.cpp:142-210) -- NOT the engine's IsAttackedByTeam: its own probe
// set, bounded by BoardLocation::Present() only (engine/board.h:194-201: inside the ARRAY), so its rays run THROUGH the
// cut corners, and the queried location may itself lie in one (GetAttackedSquaresPlayers sweeps every row x column).
bool attacked_by_player(const orc_board *b, int R, int sq, int colour) {
  const int lr = sq / R, lc = sq % R;
  static const int eight[8][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};   // board.cpp:152-153 (= :176, :201)
  // pawns (:152-161): a pawn of that colour on one of the eight neighbours that PawnAttacks the location
  // (engine/board.cpp:583-604: row_diff / col_diff = location - pawn)
  for (const auto &d : eight) {
    const int r = lr + d[0], c = lc + d[1];
    if (!in_array(R, r, c)) continue;
    const uint8_t p = b->sq[r * R + c];
    if (!present(p) || colour_of(p) != colour || type_of(p) != PAWN) continue;
    const int row_diff = lr - r, col_diff = lc - c;
    bool att = false;
    switch (colour) {
      case 0: att = row_diff == -1 && (col_diff == 1 || col_diff == -1); break;
      case 1: att = col_diff == 1 && (row_diff == 1 || row_diff == -1); break;
      case 2: att = row_diff == 1 && (col_diff == 1 || col_diff == -1); break;
      default: att = col_diff == -1 && (row_diff == 1 || row_diff == -1); break;
    }
    if (att) return true;
  }
  // knights (:164-171)
  static const int knight[8][2] = {{1, 2}, {2, 1}, {-1, -2}, {-2, -1}, {1, -2}, {2, -1}, {-1, 2}, {-2, 1}};
  for (const auto &d : knight) {
    const int r = lr + d[0], c = lc + d[1];
    if (!in_array(R, r, c)) continue;
    const uint8_t p = b->sq[r * R + c];
    if (present(p) && colour_of(p) == colour && type_of(p) == KNIGHT) return true;
  }
  // bishops, rooks, queens (:174-197): the first piece on each of the eight rays
  for (const auto &d : eight) {
    int r = lr + d[0], c = lc + d[1];
    while (in_array(R, r, c)) {
      const uint8_t p = b->sq[r * R + c];
      if (present(p)) {
        if (colour_of(p) != colour) break;
        const int t = type_of(p);
        if (t != BISHOP && t != ROOK && t != QUEEN) break;
        if (t == BISHOP && (d[0] == 0 || d[1] == 0)) break;
        if (t == ROOK && (d[0] != 0 && d[1] != 0)) break;
        return true;
      }
      r += d[0]; c += d[1];
    }
  }
  // kings (:200-207)
  for (const auto &d : eight) {
    const int r = lr + d[0], c = lc + d[1];
    if (!in_array(R, r, c)) continue;
    const uint8_t p = b->sq[r * R + c];
    if (present(p) && colour_of(p) == colour && type_of(p) == KING) return true;
  }
  return false;
}

int orc_make_move(orc_board *b, int R, const orc_move *m) { (void)R; return make_move(b, m); }
int orc_is_attacked_by_player(const orc_board *b, int R, int sq, int colour) { return attacked_by_player(b, R, sq, colour); }
// GetAttackedSquaresPlayers (board.cpp:120-140) and GetAttackedSquaresTeams (:212-232) as byte maps over EVERY row x
// column of the array (cut corners included, as the reference's double loop): out[k][sq] = 1 if attacked, k = colour
// 0..3 (IsAttackedByPlayer), then 4 + team (the engine's IsAttackedByTeam)
void orc_attack_maps(const orc_board *b, int R, int INV, uint8_t *out /* [6][R*R] */) {
  for (int sq = 0; sq < R * R; ++sq) {
    for (int colour = 0; colour < 4; ++colour) out[colour * R * R + sq] = attacked_by_player(b, R, sq, colour) ? 1 : 0;
    for (int team = 0; team < 2; ++team) out[(4 + team) * R * R + sq] = attacked_by_team(b, R, INV, team, sq) ? 1 : 0;
  }
}
int orc_take_action_flat(orc_board *b, int R, int flat) {   // board.cpp:234-239 + move.cpp:39-61
  int from, to;
  orc_flat_to_move(R, flat, &from, &to);
  orc_move m = mk_move(from, to, 0);
  if (g_rules & RULES_FULL_MOVES) {
    // the move as the generator describes it (queen promotion, rook hop, castling rights): the last
    // generated move with this (from, to) -- promotions are emitted N, B, R, Q (engine/board.cpp:82-88)
    const int INV = R == 8 || R == 10 ? 2 : 3;
    orc_move buf[300];
    orc_board tmp = *b;
    tmp.turn = (uint8_t)colour_of(b->sq[from]);
    const int n = present(b->sq[from]) ? pseudo_legal(&tmp, R, INV, buf, 300) : 0;
    for (int i = 0; i < n; ++i)
      if (buf[i].from == from && buf[i].to == to) m = buf[i];
  }
  return make_move(b, &m);
}

// engine/board.cpp:1263-1292
int orc_heuristic(const orc_board *b, int team) {
  static const int val[6] = {1, 3, 3, 5, 9, 0};
  int h = 0;
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < b->plen[c]; ++i) {
      uint8_t p = b->sq[b->pl[c][i]];
      if (!present(p) || type_of(p) == KING) continue;
      h += team_of(p) == team ? val[type_of(p)] : -val[type_of(p)];
    }
  return h;
}

int orc_rot90_src(int R, int k, int i, int j) { return rot90_src(R, k, i, j); }

// board.cpp:285-296, :305-356
void orc_encode(const orc_board *boards, int n, int R, float *out) {
  int RR = R * R;
  std::vector<float> tmp((size_t)24 * RR);
  int k = n > 0 ? boards[0].turn : 0;  // batch-wide rotation by states[0] (quirk Q6)
  for (int bi = 0; bi < n; ++bi) {
    const orc_board *b = &boards[bi];
    if (g_rules & RULES_ROTATION) k = b->turn;
    std::fill(tmp.begin(), tmp.end(), 0.f);
    for (int c = 0; c < 4; ++c)
      for (int i = 0; i < b->plen[c]; ++i) {
        int sq = b->pl[c][i];
        uint8_t p = b->sq[sq];
        int off = 6 * ((colour_of(p) - b->turn + 4) & 3);
        int plane = off + type_of(p) - 1;    // :336; -1 wraps to plane 23 (index_put_ negative index, Q7)
        if (g_rules & RULES_PLANES) plane = off + type_of(p);
        if (plane < 0) plane += 24;
        tmp[(size_t)plane * RR + sq] = 1.f;
      }
    float *o = out + (size_t)bi * 24 * RR;
    for (int p = 0; p < 24; ++p)
      for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) o[(size_t)p * RR + i * R + j] = tmp[(size_t)p * RR + rot90_src(R, k, i, j)];
  }
}

// ---- deterministic exp (shared numeric spec, DESIGN.md "fpc_expf") ----
float orc_expf(float x) {
  if (x != x) return x;
  if (x < -86.0f) return 0.0f;
  if (x > 88.0f) return std::numeric_limits<float>::infinity();
  float k = __builtin_rintf(x * 0x1.715476p+0f);
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
  int ki = (int)k;
  uint32_t bits = (uint32_t)(ki + 127) << 23;
  float s;
  memcpy(&s, &bits, 4);
  return p * s;
}

// mcts.py:67-76.  Numeric spec ("policy head-to-prior arithmetic"):
//   the row is taken in chunks of 1024 logits (256 "threads" x float4).  Chunk c: m_c = its maximum;
//   s_c = sum of fpc_expf(l - m_c) where thread t owns float4 group t of the chunk and adds x,y,z,w in
//   order, the 64 partials of each of the four "waves" are folded by the xor butterfly 32,16,..,1 and the
//   wave sums are added ((w0+w1)+w2)+w3 (a chunk that is all -inf has s_c = 0);
//   m = max_c m_c;  S = sequential ascending sum of s_c * fpc_expf(m_c - m) (chunks with m_c = -inf add 0);
//   p_i = fpc_expf(l_i - m) * (1/S); priors = p_src / (sequential ascending sum of p over the legal set).
int orc_policy_priors(const float *logits, int R, int turn0, const int *legal_flat, int n_legal, float *priors) {
  int A = orc_action_size(R), RR = R * R;
  const float NINF = -std::numeric_limits<float>::infinity();
  bool has_nan = false;
  const int ngroups = A / 4, nchunks = (ngroups + 255) / 256;
  std::vector<float> mc(nchunks), sc(nchunks);
  for (int c = 0; c < nchunks; ++c) {
    float m = NINF;
    for (int t = 0; t < 256; ++t) {
      const int q = c * 256 + t;
      if (q >= ngroups) continue;
      for (int k = 0; k < 4; ++k) { const float v = logits[4 * q + k]; if (v != v) has_nan = true; if (v > m) m = v; }
    }
    float part[256];
    for (int t = 0; t < 256; ++t) {
      const int q = c * 256 + t;
      part[t] = 0.f;
      if (q < ngroups && m > NINF)
        for (int k = 0; k < 4; ++k) part[t] = part[t] + orc_expf(logits[4 * q + k] - m);
    }
    for (int w = 0; w < 4; ++w)
      for (int off = 32; off >= 1; off >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = part[w * 64 + l] + part[w * 64 + (l ^ off)];
        memcpy(part + w * 64, nxt, sizeof(nxt));
      }
    mc[c] = m;
    sc[c] = ((part[0] + part[64]) + part[128]) + part[192];
  }
  float m = NINF;
  for (int c = 0; c < nchunks; ++c) if (mc[c] > m) m = mc[c];
  float S = 0.f;
  for (int c = 0; c < nchunks; ++c) S = S + (mc[c] > NINF ? sc[c] * orc_expf(mc[c] - m) : 0.f);
  float inv = 1.0f / S;
  float T = 0.f;
  for (int j = 0; j < n_legal; ++j) {
    int plane = legal_flat[j] / RR, pos = legal_flat[j] % RR;
    int src = plane * RR + rot90_src(R, -turn0, pos / R, pos % R);
    priors[j] = orc_expf(logits[src] - m) * inv;
    T = T + priors[j];
  }
  if (has_nan || !(T > 0.f) || T != T) return 1;
  for (int j = 0; j < n_legal; ++j) priors[j] = priors[j] / T;
  return 0;
}

// ---- MCTS: node.h:70-78, node.cpp, mcts.py ----
namespace {
struct ONode {
  int visit_count;       // node.h:70
  double value_sum;      // node.h:76
  double prior;          // node.h:75
  int parent;            // node.h:73
  int flat;              // move_made (GetFlatIndex)
  std::vector<int> children;
  orc_board state;       // every node owns a Board copy (node.h:72, node.cpp:90-91)
  bool state_is_root;    // root shares the caller's Board (mcts.py:30)
};
}

int orc_search(orc_board *boards, int G, int R, int INV, int sims, double Cpuct, orc_eval_fn eval, void *user,
               orc_search_out *out, int max_children, int *child_flat, int *child_visits, float *child_prior,
               double *child_w) {
  const int A = orc_action_size(R), RR = R * R;
  std::vector<std::vector<ONode>> trees(G);
  std::vector<char> alive(G, 1);
  std::vector<int> sims_done(G, 0);
  for (int g = 0; g < G; ++g) {
    trees[g].reserve(1024);
    ONode root{};
    root.visit_count = 1; root.value_sum = 0; root.prior = 0; root.parent = -1; root.flat = -1;
    root.state = boards[g]; root.state_is_root = true;
    trees[g].push_back(root);
  }
  auto backprop = [&](std::vector<ONode> &t, int n, float v) {  // node.cpp:133-142
    while (n >= 0) { t[n].value_sum += v; t[n].visit_count += 1; v = -v; n = t[n].parent; }
  };
  std::vector<float> enc, logits, value;
  std::vector<orc_board> leaf_states;
  std::vector<int> leaf_game, leaf_node;
  int rc = 0;
  for (int s = 0; s < sims; ++s) {
    leaf_states.clear(); leaf_game.clear(); leaf_node.clear();
    for (int g = 0; g < G; ++g) {            // mcts.py:18-26
      if (!alive[g]) continue;
      auto &t = trees[g];
      int n = 0;
      while (!t[n].children.empty()) {       // node.cpp:23-26, :49-78
        const ONode &nd = t[n];
        int best = -1;
        double best_ucb = -std::numeric_limits<double>::infinity();
        double lp = std::log(std::sqrt((double)nd.visit_count));
        for (size_t i = 0; i < nd.children.size(); ++i) {
          const ONode &ch = t[nd.children[i]];
          double cv = ch.visit_count > 0 ? ch.value_sum / ch.visit_count : 0;
          double ucb = cv + Cpuct * std::sqrt(lp / (1 + ch.visit_count)) * ch.prior;
          if (g_rules & RULES_PUCT) {
            const double q = ch.visit_count > 0 ? -(ch.value_sum / (double)ch.visit_count) : 0.0;
            ucb = q + Cpuct * ch.prior * std::sqrt((double)nd.visit_count) / (double)(1 + ch.visit_count);
          }
          if (ucb > best_ucb) { best = (int)i; best_ucb = ucb; }
        }
        if (best < 0) { rc = -2; goto done; }  // node.cpp:72-75 throws
        n = nd.children[best];
      }
      int res = game_result(&t[n].state, R, INV, -1);   // node.cpp:28-29
      if (res != IN_PROGRESS) {                         // node.cpp:31-42, quirk Q5
        backprop(t, n, res == STALEMATE ? 0.f : -1.f);
        sims_done[g]++;
        alive[g] = 0;
        continue;
      }
      leaf_game.push_back(g); leaf_node.push_back(n);
    }
    int B = (int)leaf_game.size();
    if (B == 0) continue;                               // mcts.py:60-61
    leaf_states.resize(B);
    for (int i = 0; i < B; ++i) leaf_states[i] = trees[leaf_game[i]][leaf_node[i]].state;
    enc.resize((size_t)B * 24 * RR); logits.resize((size_t)B * A); value.resize(B);
    orc_encode(leaf_states.data(), B, R, enc.data());   // mcts.py:65
    eval(user, enc.data(), B, logits.data(), value.data());  // mcts.py:66
    int turn0 = leaf_states[0].turn;                    // mcts.py:69 (quirk Q6)
    // legal masks (four_player_chess_board.py:36-56): GetLegalMoves mutates each leaf state
    std::vector<std::vector<int>> legal(B);
    std::vector<std::vector<float>> pri(B);
    for (int i = 0; i < B; ++i) {
      orc_board &st = trees[leaf_game[i]][leaf_node[i]].state;
      orc_move mv[300];
      int n = legal_moves(&st, R, INV, mv, 300);
      for (int k = 0; k < n; ++k) legal[i].push_back(move_plane(R, mv[k].from, mv[k].to) * RR + mv[k].from);
      std::sort(legal[i].begin(), legal[i].end());
      legal[i].erase(std::unique(legal[i].begin(), legal[i].end()), legal[i].end());
      pri[i].resize(legal[i].size());
      const int rot = (g_rules & RULES_ROTATION) ? leaf_states[i].turn : turn0;
      if (orc_policy_priors(&logits[(size_t)i * A], R, rot, legal[i].data(), (int)legal[i].size(), pri[i].data())) {
        rc = -3; goto done;
      }
      if (g_noise && leaf_node[i] == 0) {              // root Dirichlet noise (N4), same arithmetic as the engine
        const float *gm = g_noise + (size_t)leaf_game[i] * g_noise_stride;
        float sg = 0.f;
        for (size_t k2 = 0; k2 < legal[i].size(); ++k2) sg = sg + gm[k2];
        if (sg > 0.f) for (size_t k2 = 0; k2 < legal[i].size(); ++k2) pri[i][k2] = (1.0f - g_noise_eps) * pri[i][k2] + g_noise_eps * (gm[k2] / sg);
      }
    }
    for (int i = 0; i < B; ++i) {                       // mcts.py:78, node.cpp:144-154
      backprop(trees[leaf_game[i]], leaf_node[i], value[i]);
      sims_done[leaf_game[i]]++;
    }
    for (int i = 0; i < B; ++i) {                       // mcts.py:79-89, node.cpp:79-131
      auto &t = trees[leaf_game[i]];
      int n = leaf_node[i];
      for (size_t k = 0; k < legal[i].size(); ++k) {
        if (pri[i][k] == 0.f) continue;                 // torch.nonzero drops exact zeros (Q: underflow)
        ONode ch{};
        ch.visit_count = 1;                             // node.h:28 default (quirk Q1)
        ch.value_sum = 0; ch.prior = (double)pri[i][k]; ch.parent = n; ch.flat = legal[i][k];
        ch.state = t[n].state; ch.state_is_root = false;
        if (orc_take_action_flat(&ch.state, R, ch.flat) != 0) { rc = -4; goto done; }
        t.push_back(ch);
        t[n].children.push_back((int)t.size() - 1);
      }
    }
  }
done:
  for (int g = 0; g < G; ++g) {
    auto &t = trees[g];
    boards[g] = t[0].state;   // root shares the caller's Board object
    out[g].root_visits = t[0].visit_count;
    out[g].n_children = (int)t[0].children.size();
    out[g].terminated = !alive[g];
    out[g].sims_done = sims_done[g];
    for (int k = 0; k < (int)t[0].children.size() && k < max_children; ++k) {
      const ONode &ch = t[t[0].children[k]];
      child_flat[(size_t)g * max_children + k] = ch.flat;
      child_visits[(size_t)g * max_children + k] = ch.visit_count;
      if (child_prior) child_prior[(size_t)g * max_children + k] = (float)ch.prior;
      if (child_w) child_w[(size_t)g * max_children + k] = ch.value_sum;
    }
  }
  return rc;
}

void orc_eval_zero(void *user, const float *enc, int B, float *logits, float *value) {
  (void)enc;
  int R = ((orc_eval_ctx *)user)->R, A = orc_action_size(R);
  for (size_t i = 0; i < (size_t)B * A; ++i) logits[i] = 0.f;
  for (int b = 0; b < B; ++b) value[b] = 0.f;
}
// SURVEY.md section 4 "ramp": logits[b,i] = -(i mod 7)/8 ; value[b] = ((sum x*w) mod 5 - 2)/4,
// w = arange(24*R*R) mod 11
void orc_eval_ramp(void *user, const float *enc, int B, float *logits, float *value) {
  int R = ((orc_eval_ctx *)user)->R, A = orc_action_size(R), S = 24 * R * R;
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < A; ++i) logits[(size_t)b * A + i] = -(float)(i % 7) / 8.0f;
    double acc = 0;
    for (int i = 0; i < S; ++i) acc += enc[(size_t)b * S + i] * (double)(i % 11);
    long v = (long)acc;
    value[b] = (float)((double)(v % 5) - 2.0) / 4.0f;
  }
}

}  // extern "C"
