"""Batched self-play episodes -- the engine-side counterpart of AlphaZero.play()
(/root/reference/src/py/alphazero.py:81-178) and handle_terminal_state (:53-78).

Semantics kept from the reference:
  * one MCTS.search per ply over the still-running games, in list order (finished games are deleted
    from the list, so the batch -- and with it the Q6 batch-wide rotation -- shrinks exactly as there);
  * pi[flat] = child visit count / sum (f32), temperature pow(pi, 1/T) renormalised (f32) (:104-116);
  * the move is drawn from those probabilities (:118).  The reference uses an unseeded
    torch.multinomial; here the draw is an explicit inverse-CDF over ascending flat index driven by a
    caller-supplied uniform, so that traces are reproducible (`sample_action`);
  * TakeAction(Move(flat)) -> GetGameResult (:119-123);
  * terminal: the team that just MOVED is `losing_team`, every stored tuple of that game gets
    z = +1 if its side-to-move team != losing_team else -1 (:128-137, quirk Q12);
  * games alive after max_game_length plies are scored by the material heuristic of the side to move
    times heuristic_weight, sign by team (:161-175).
Returned tuples are compact: (root position POD, flat[], visits[], z); dense tensors are rebuilt with
tuples.dense_pi / engine.encode when the trainer needs them.
"""
import numpy as np

import fpc_ffi


def sample_action(flats, visits, temperature, u):
    """alphazero.py:104-119 with an explicit uniform u in [0,1)."""
    p = np.asarray(visits, dtype=np.float32)
    p = p / p.sum(dtype=np.float32)
    t = np.power(p, np.float32(1.0 / temperature), dtype=np.float32)
    t = t / t.sum(dtype=np.float32)
    c = np.cumsum(t.astype(np.float64))
    k = int(np.searchsorted(c, u * c[-1], side="right"))
    return int(flats[min(k, len(flats) - 1)])


class Episode:
    def __init__(self, gid):
        self.gid = gid
        self.entries = []       # (board POD snapshot, flats, visits)
        self.moves = []
        self.z = []             # per entry
        self.result = 0
        self.length = 0


def play(search_fn, eng, start_boards, args, uniforms):
    """search_fn(list_of_PODs) -> search_results dict (fpc_ffi.Engine.search_results layout) and
    leaves the PODs with the piece-list order the search produced.  uniforms[ply][gid] in [0,1).
    Returns the list of finished Episodes (all games, in game-id order)."""
    R = eng.R
    states = [fpc_ffi.clone_board(b) for b in start_boards]
    ids = list(range(len(states)))
    eps = {g: Episode(g) for g in ids}
    T = float(args["temperature"])
    for ply in range(int(args["max_game_length"])):
        if not states:
            break
        res = search_fn(states)
        picks = []
        for i in range(len(states)):
            n = int(res["n_children"][i])
            flats, visits = res["flat"][i, :n].copy(), res["visits"][i, :n].copy()
            eps[ids[i]].entries.append((fpc_ffi.clone_board(states[i]), flats, visits))
            picks.append(sample_action(flats, visits, T, uniforms[ply][ids[i]]))
        nxt = eng.take_action(states, picks)
        results = eng.game_result(nxt)
        keep_s, keep_i = [], []
        for i in range(len(states)):
            e = eps[ids[i]]
            e.moves.append(picks[i])
            e.length += 1
            if results[i] != 0:
                e.result = int(results[i])
                losing_team = states[i].turn & 1               # team of the player who just moved (Q12)
                e.z = [1.0 if (b.turn & 1) != losing_team else -1.0 for b, _, _ in e.entries]
            else:
                keep_s.append(nxt[i]); keep_i.append(ids[i])
        states, ids = keep_s, keep_i
    for s, g in zip(states, ids):                                # max_game_length reached (:161-175)
        e = eps[g]
        curr_team = s.turn & 1
        h = eng.L.fpc_board_heuristic(s, curr_team) * float(args["heuristic_weight"])
        e.z = [h if (b.turn & 1) == curr_team else -h for b, _, _ in e.entries]
    return [eps[g] for g in sorted(eps)]
