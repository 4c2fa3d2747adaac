"""Arena evaluation: paired games between two evaluators at temperature 0 (BASELINE.json
configs[4]: "paired games new-vs-reference weights, temperature=0, bit-exact legal-move/visit-count
check").

The reference has no arena and its move selection cannot take T = 0 (pow(pi, 1/T),
/root/reference/src/py/alphazero.py:104-119), so this harness is ours; what it
keeps from the reference is everything underneath: MCTS.search per ply over the games whose side to
move belongs to an evaluator (mcts.py:17-43, with its batch-wide rotation by the first leaf, quirk
Q6), pi = child visit counts (alphazero.py:104-110), TakeAction(Move(flat)) and GetGameResult
(:119-123), and the material heuristic for games that reach max_game_length (:161-175).

  * T -> 0 limit of the temperature rule: the most visited root child; ties -> the lowest flat index
    (children are reported in ascending flat order, so the first maximum).
  * Pairing: every start position is played twice, evaluator A owning team RED/YELLOW in the first
    game and BLUE/GREEN in the second, so a first-move advantage cancels.
  * One ply = two batched searches: A's over the live games where A is to move (in game order), then
    B's over the rest.  Each batch is what MCTS.search sees, so results depend on the batch
    composition exactly as the reference's would (Q6).
  * Result of a game: GameResult WIN_RY / WIN_BG decide it, STALEMATE is a draw; a game still running
    after max_game_length plies is adjudicated by the sign of CalculateHeuristic for the side to move.
"""
import numpy as np

import fpc_ffi

WIN_RY, WIN_BG, STALEMATE = 1, 2, 3


def pick_argmax(flats, visits):
    return int(flats[int(np.argmax(np.asarray(visits)))])


class ArenaGame:
    def __init__(self, gid, pair, a_team, state):
        self.gid, self.pair, self.a_team = gid, pair, a_team
        self.state = state
        self.plies = []          # (turn, flats[], visits[], pick)
        self.result = 0          # GameResult when the game ended on the board
        self.winner = None       # 0 = RY, 1 = BG, -1 = draw

    def score_a(self):
        return 0.5 if self.winner < 0 else (1.0 if self.winner == self.a_team else 0.0)


def play_paired(search_a, search_b, eng, start_boards, args):
    """search_x(list_of_PODs) -> search_results dict (fpc_ffi.Engine.search_results layout), leaving
    the PODs with the piece-list order the search produced.  Returns the list of ArenaGames."""
    games = []
    for k, b in enumerate(start_boards):
        for a_team in (0, 1):
            games.append(ArenaGame(len(games), k, a_team, fpc_ffi.clone_board(b)))
    live = list(games)
    for _ply in range(int(args["max_game_length"])):
        if not live:
            break
        for fn, batch in ((search_a, [g for g in live if (g.state.turn & 1) == g.a_team]),
                          (search_b, [g for g in live if (g.state.turn & 1) != g.a_team])):
            if not batch:
                continue
            pods = [g.state for g in batch]
            res = fn(pods)
            picks = []
            for i, g in enumerate(batch):
                n = int(res["n_children"][i])
                flats, visits = res["flat"][i, :n].copy(), res["visits"][i, :n].copy()
                picks.append(pick_argmax(flats, visits))
                g.plies.append((int(pods[i].turn), flats, visits, picks[-1]))
            nxt = eng.take_action(pods, picks)
            results = eng.game_result(nxt)
            for i, g in enumerate(batch):
                g.state = nxt[i]
                if results[i] != 0:
                    g.result = int(results[i])
                    g.winner = 0 if g.result == WIN_RY else (1 if g.result == WIN_BG else -1)
        live = [g for g in live if g.result == 0]
    for g in live:                                    # max_game_length reached
        team = g.state.turn & 1
        h = eng.L.fpc_board_heuristic(g.state, team)
        g.winner = -1 if h == 0 else (team if h > 0 else 1 - team)
    return games


def summary(games):
    s = sum(g.score_a() for g in games)
    return {"games": len(games), "score_a": s, "score_b": len(games) - s,
            "wins_a": sum(1 for g in games if g.winner == g.a_team),
            "wins_b": sum(1 for g in games if g.winner == 1 - g.a_team),
            "draws": sum(1 for g in games if g.winner < 0),
            "plies": sum(len(g.plies) for g in games)}


def nn_search_fn(eng, sims, c_puct):
    """search over the engine's internal MFMA ResNet (weights already loaded into `eng`)."""
    def fn(pods):
        eng.search_begin(pods, c_puct)
        eng.search_run(sims)
        return eng.search_results(roots=pods)
    return fn
