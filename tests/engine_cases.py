"""Parity cases shared by tests/test_engine_emul.py (CPU, wavefront emulator) and
tests/test_engine_gpu.py (real MI355X, through libfpc_engine.so).  Every case compares the HIP
kernels' results with (a) the golden vectors from the real reference and/or (b) the CPU oracle on
the same inputs -- bit-exact (integer/byte/index work; visit counts; f32 priors; f64 value sums)."""
import numpy as np

import evaluators
import fpc_ffi
from fpc_testlib import expand_promos, gold, make_engine, run_external_search
from oracle import orc


def _orc_board(fb, R):
    return orc.board_from_lists(R, fb.turn, fpc_ffi.lists_of(fb))


def case_static(backend, R):
    g = gold(R)
    L = fpc_ffi.lib() if backend == "gpu" else __import__("fpc_testlib").emul_lib()
    assert L.fpc_num_action_channels(R) == g["A_ch"] and L.fpc_action_space_size(R) == g["A"]
    for r in range(R):
        for c in range(R):
            assert L.fpc_is_legal_location(R, g["INV"], r, c) == g["legal_loc"][r][c]
    st = g["start"]
    b = fpc_ffi.board_from_dict(R, st["turn"], [tuple(e) for e in st["dict"]], _lib=L)
    assert fpc_ffi.lists_of(b) == st["after_ctor"]["pl"]      # constructor order (engine/board.cpp:1209-1247)
    import ctypes as C
    for flat, frm, to in g["codec_flat_to_move"]:
        if flat // (R * R) >= 8 * (R - 1) + 8:
            continue
        f, t = C.c_int(), C.c_int()
        assert L.fpc_flat_to_move(R, flat, C.byref(f), C.byref(t)) == 0
        assert f.value == frm
        if t.value == fpc_ffi.NO_SQ:
            assert to == R * R
        else:
            assert t.value == to and L.fpc_move_flat_index(R, frm, to) == flat


def case_playouts(backend, R, max_games, max_plies):
    """Lock-step replay of the recorded reference games, all games batched in one launch per call:
    GetGameResult -> GetLegalMoves -> encode -> TakeAction, comparing results and piece-list order."""
    g = gold(R)
    INV = g["INV"]
    eng = make_engine(backend, R, INV, max_games=4, max_sims=4)
    games = g["playouts"][:max_games]
    boards = [fpc_ffi.board_from_lists(R, gm[0]["before"]["turn"], gm[0]["before"]["pl"]) for gm in games]
    live = list(range(len(games)))
    n_pos = n_term = n_child = 0
    for ply in range(max_plies):
        live = [i for i in live if ply < len(games[i])]
        if not live:
            break
        recs = [games[i][ply] for i in live]
        bs = [boards[i] for i in live]
        for b, rec in zip(bs, recs):
            assert b.turn == rec["before"]["turn"] and fpc_ffi.lists_of(b) == rec["before"]["pl"]
        res = eng.game_result(bs)
        for b, rec, r in zip(bs, recs, res):
            assert r == rec["result"], (ply, r, rec["result"])
            assert fpc_ffi.lists_of(b) == rec["after_result"]
            n_pos += 1
        cont = [k for k, rec in enumerate(recs) if rec["result"] == 0]
        n_term += len(recs) - len(cont)
        if not cont:
            live = []
            break
        bs = [bs[k] for k in cont]
        recs = [recs[k] for k in cont]
        live = [live[k] for k in cont]
        lm = eng.legal_moves(bs)
        for b, rec, m in zip(bs, recs, lm):
            assert expand_promos(m) == rec["legal"], ply
            assert fpc_ffi.lists_of(b) == rec["after_legal"]
        encs = [k for k, rec in enumerate(recs) if "enc" in rec]
        for k in encs:
            e = eng.encode([bs[k]])
            assert np.nonzero(e.flatten())[0].tolist() == recs[k]["enc"]
        for b, rec in zip(bs, recs):
            if "children" in rec and ply % 7 == 0:
                flats = [c[0] for c in rec["children"]]
                outs = eng.take_action([b] * len(flats), flats)
                for nb, (fl, snap) in zip(outs, rec["children"]):
                    assert nb.turn == snap["turn"] and fpc_ffi.lists_of(nb) == snap["pl"]
                    n_child += 1
        nxt = eng.take_action(bs, [rec["pick"] for rec in recs])
        for i, nb in zip(live, nxt):
            boards[i] = nb
    eng.close()
    return n_pos, n_term, n_child


def case_batch_encode(backend, R):
    g = gold(R)
    be = g["batch_encode"]
    eng = make_engine(backend, R, g["INV"], max_games=4, max_sims=4)
    boards = [fpc_ffi.board_from_lists(R, s["turn"], s["pl"]) for s in be["states"]]
    e = eng.encode(boards)
    assert list(e.shape) == be["shape"]
    assert np.nonzero(e.flatten())[0].tolist() == be["enc"]
    # legal mask == dense form of the legal list (four_player_chess_board.py:36-56)
    m = eng.legal_mask([fpc_ffi.clone_board(b) for b in boards])
    lm = eng.legal_moves([fpc_ffi.clone_board(b) for b in boards])
    for i in range(len(boards)):
        assert sorted(set(x[2] for x in lm[i])) == np.nonzero(m[i].flatten())[0].tolist()
    eng.close()


def _compare_search(res, oref, tag):
    """engine result dict vs oracle result list (bit-exact, including f32 priors and f64 W)."""
    for gi, o in enumerate(oref):
        n = int(res["n_children"][gi])
        assert int(res["root_n"][gi]) == o["root_n"], tag
        got = [[int(res["flat"][gi, k]), int(res["visits"][gi, k])] for k in range(n)]
        assert got == o["children"], (tag, gi)
        assert int(res["sims_done"][gi]) == o["sims_done"], tag
        assert np.array_equal(res["prior"][gi, :n], o["priors"]), (tag, gi, "priors")
        assert np.array_equal(res["w"][gi, :n], o["w"]), (tag, gi, "value sums")
        assert fpc_ffi.lists_of(res["boards"][gi]) == orc.lists_of(o["board"]), (tag, gi, "root list order")


def case_search_golden(backend, R, max_sims, kinds=None, max_cases=None):
    """MCTS.search visit counts (root + second level) vs the golden vectors of the real reference and
    vs the oracle."""
    g = gold(R)
    INV = g["INV"]
    done = 0
    for si, rec in enumerate(g["searches"]):
        if rec["sims"] > max_sims or (kinds and rec["kind"] not in kinds):
            continue
        if max_cases is not None and done >= max_cases:
            break
        done += 1
        ev = evaluators.make(rec["kind"], R)
        roots = [fpc_ffi.board_from_lists(R, s["turn"], s["pl"]) for s in rec["before"]]
        eng = make_engine(backend, R, INV, max_games=len(roots), max_sims=rec["sims"])
        # every other case drives the loop through the fused fpc_search_expand_select entry point
        res = run_external_search(eng, backend, roots, rec["sims"], rec["C"], ev, fused=bool(done & 1))
        tag = (si, rec["kind"], rec["sims"], "fused" if done & 1 else "stepwise")
        for gi, ref in enumerate(rec["roots"]):
            n = int(res["n_children"][gi])
            got = [[int(res["flat"][gi, k]), int(res["visits"][gi, k])] for k in range(n)]
            assert int(res["root_n"][gi]) == ref["root_n"], tag
            assert got == [[c[0], c[1]] for c in ref["children"]], (tag, gi)
            assert fpc_ffi.lists_of(roots[gi]) == ref["after"], (tag, gi)
            for k, c in enumerate(ref["children"]):
                if len(c) > 2 and (k % 5 == 0):
                    assert eng.grandchildren(gi, k) == c[2], (tag, gi, k)
        oboards = [orc.board_from_lists(R, s["turn"], s["pl"]) for s in rec["before"]]
        rc, oref = orc.search(oboards, R, INV, rec["sims"], rec["C"], rec["kind"] if rec["kind"] in ("zero", "ramp") else ev)
        assert rc == 0
        _compare_search(res, oref, tag)
        eng.close()
    return done


def case_search_random_vs_oracle(backend, R, n_games, sims, seed, kind="hash"):
    """Seeded mid-game positions (random playouts through the ORACLE), searched by both sides."""
    g = gold(R)
    INV = g["INV"]
    import random
    rng = random.Random(seed)
    st = g["start"]
    roots_o = []
    for _ in range(n_games):
        b = orc.board_from_dict(R, st["turn"], st["dict"])
        for _ply in range(rng.randrange(0, 40)):
            if orc.game_result(b, R, INV) != 0:
                break
            lm = orc.legal_moves(b, R, INV)
            flats = sorted(set(x[2] for x in lm))
            nb, rc = orc.take_action(b, R, flats[rng.randrange(len(flats))])
            assert rc == 0
            b = nb
        if orc.game_result(orc.clone(b), R, INV) != 0:
            b = orc.board_from_dict(R, st["turn"], st["dict"])
        roots_o.append(b)
    roots = [fpc_ffi.board_from_lists(R, b.turn, orc.lists_of(b)) for b in roots_o]
    ev = evaluators.make(kind, R)
    rc, oref = orc.search([orc.clone(b) for b in roots_o], R, INV, sims, 3.0, ev)
    eng = make_engine(backend, R, INV, max_games=n_games, max_sims=sims)
    if rc == -3:
        # some leaf had every legal move at probability 0: the reference expands all A indices and
        # throws "piece missing" (mcts.py:76,84; engine/board.cpp:1046); the engine must refuse too
        import pytest
        with pytest.raises(RuntimeError, match="policy mass"):
            run_external_search(eng, backend, roots, sims, 3.0, ev)
        eng.close()
        return "policy-error"
    assert rc == 0
    res = run_external_search(eng, backend, roots, sims, 3.0, ev)
    _compare_search(res, oref, ("random", R, seed))
    eng.close()
    return "ok"


def case_selfplay_trace(backend, R, max_traces=None):
    """Whole self-play episodes (the play() loop of alphazero.py:81-178 around MCTS.search,
    TakeAction, GetGameResult, CalculateHeuristic) against traces recorded from the real reference:
    every move, every pi (child visit counts), results, z targets (quirk Q12) and final positions."""
    import selfplay
    g = gold(R)
    INV = g["INV"]
    st = g["start"]
    n_checked = 0
    for ti, tr in enumerate(g["selfplay"][:max_traces]):
        ev = evaluators.make(tr["kind"], R)
        n = tr["n_games"]
        eng = make_engine(backend, R, INV, max_games=n, max_sims=tr["sims"])
        start = [fpc_ffi.board_from_dict(R, st["turn"], [tuple(e) for e in st["dict"]], _lib=eng.L) for _ in range(n)]
        args = {"temperature": tr["temperature"], "max_game_length": tr["max_len"], "heuristic_weight": tr["heuristic_weight"]}

        def search_fn(pods):
            return run_external_search(eng, backend, pods, tr["sims"], 3.0, ev)

        eps = selfplay.play(search_fn, eng, start, args, tr["uniforms"])
        for e, ref in zip(eps, tr["games"]):
            assert e.moves == ref["moves"], (ti, e.gid)
            assert e.result == ref["result"], (ti, e.gid)
            assert [[[int(f), int(v)] for f, v in zip(fl, vi)] for _, fl, vi in e.entries] == ref["pi"], (ti, e.gid)
            assert [b.turn for b, _, _ in e.entries] == ref["turns"]
            assert [float(z) for z in e.z] == [float(z) for z in ref["z"]], (ti, e.gid, e.z[:4], ref["z"][:4])
            n_checked += len(e.moves)
        eng.close()
    return n_checked


def synthetic_entries(R, INV):
    """A start layout for the board sizes the reference has no FEN for (9, 11, 12 a side; fpc_create accepts
    8..14): each colour gets king, queen, two rooks, two knights, a bishop on its back rank inside the cross arm
    and pawns on its second rank (the double-step line), Red bottom / Blue left / Yellow top / Green right as in
    start_fens.py.  -> (turn, [(sq, colour, type), ...]) in row-major order."""
    PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = 0, 1, 2, 3, 4, 5
    w = R - 2 * INV                                   # files of an arm
    back = [ROOK, KNIGHT, BISHOP, QUEEN, KING, BISHOP, KNIGHT, ROOK]
    back = (back[:w // 2] + back[len(back) - (w - w // 2):]) if w < 8 else back + [KNIGHT] * (w - 8)
    if KING not in back:
        back[len(back) // 2] = KING
    cells = {}
    for i in range(w):
        f = INV + i
        cells[(R - 1, f)] = (0, back[i]); cells[(R - 2, f)] = (0, PAWN)        # Red: bottom, moves up
        cells[(f, 0)] = (1, back[i]); cells[(f, 1)] = (1, PAWN)                # Blue: left, moves right
        cells[(0, f)] = (2, back[w - 1 - i]); cells[(1, f)] = (2, PAWN)        # Yellow: top, moves down
        cells[(f, R - 1)] = (3, back[w - 1 - i]); cells[(f, R - 2)] = (3, PAWN)  # Green: right, moves left
    return 0, [(r * R + c, col, typ) for (r, c), (col, typ) in sorted(cells.items())]


def case_other_sizes_vs_oracle(backend, R, INV, n_games=6, sims=40, seed=5):
    """10x10/2 and 13x13/3 (the reference's other start layouts, start_fens.py:18-56) and 9x9/2, 11x11/3,
    12x12/3 (sizes fpc_create accepts without a reference layout: synthetic_entries): no compiled reference
    exists for them, so the engine is compared with the oracle (pinned at 8 and 14)."""
    import random
    import positions
    turn, entries = positions.start_entries(R) if R in (8, 10, 13, 14) else synthetic_entries(R, INV)
    rng = random.Random(seed)
    eng = make_engine(backend, R, INV, max_games=n_games, max_sims=sims)
    roots_o = []
    for _ in range(n_games):
        b = orc.board_from_dict(R, turn, [list(e) for e in entries])
        fb = fpc_ffi.board_from_dict(R, turn, entries, _lib=eng.L)
        assert fpc_ffi.lists_of(fb) == orc.lists_of(b)
        for _ply in range(rng.randrange(0, 30)):
            r_e, r_o = eng.game_result([fb])[0], orc.game_result(b, R, INV)
            assert r_e == r_o
            if r_o != 0:
                break
            lm_e = expand_promos(eng.legal_moves([fb])[0])
            lm_o = orc.legal_moves(b, R, INV)
            assert lm_e == lm_o and fpc_ffi.lists_of(fb) == orc.lists_of(b)
            flats = sorted(set(x[2] for x in lm_o))
            pick = flats[rng.randrange(len(flats))]
            b, rc = orc.take_action(b, R, pick)
            fb = eng.take_action([fb], [pick])[0]
            assert rc == 0 and fpc_ffi.lists_of(fb) == orc.lists_of(b)
        if orc.game_result(orc.clone(b), R, INV) != 0:
            b = orc.board_from_dict(R, turn, [list(e) for e in entries])
        roots_o.append(b)
    ev = evaluators.make("hash", R)
    rc, oref = orc.search([orc.clone(b) for b in roots_o], R, INV, sims, 3.0, ev)
    assert rc == 0
    roots = [fpc_ffi.board_from_lists(R, b.turn, orc.lists_of(b)) for b in roots_o]
    res = run_external_search(eng, backend, roots, sims, 3.0, ev)
    _compare_search(res, oref, ("size", R))
    eng.close()
    return True


def case_castling_vs_oracle(backend, n_games=4, plies=40, sims=30, seed=21):
    """Castling (engine/board.cpp:343-465).  NOT reachable through the reference's Python boundary:
    `Board(turn, l2p, castling_rights)` needs a dict keyed by Player, and the bound Player type is
    unhashable (wrapper.cpp:81-87 defines == without __hash__), while fen_parser drops the rights
    (Q10) -- so no golden vector can exist.  The C-ABI accepts rights; the device implementation is
    held against the oracle's restatement: 14x14 start without knights/bishops/queens, all rights set."""
    import random
    import positions
    R, INV = 14, 3
    turn, entries = positions.start_entries(R)
    entries = [e for e in entries if e[2] in (positions.PAWN, positions.ROOK, positions.KING)]
    rng = random.Random(seed)
    eng = make_engine(backend, R, INV, max_games=n_games, max_sims=sims)
    n_castle = 0
    roots_o = []
    for _ in range(n_games):
        b = orc.board_from_dict(R, turn, [list(e) for e in entries], castle=[3, 3, 3, 3])
        fb = fpc_ffi.board_from_dict(R, turn, entries, castle=[3, 3, 3, 3], _lib=eng.L)
        assert fpc_ffi.lists_of(fb) == orc.lists_of(b)
        for _ply in range(plies):
            r_e, r_o = eng.game_result([fb])[0], orc.game_result(b, R, INV)
            assert r_e == r_o and fpc_ffi.lists_of(fb) == orc.lists_of(b)
            if r_o != 0:
                break
            lm_e = expand_promos(eng.legal_moves([fb])[0])
            lm_o = orc.legal_moves(b, R, INV)
            assert lm_e == lm_o, (_ply, lm_e, lm_o)
            assert fpc_ffi.lists_of(fb) == orc.lists_of(b)
            kings = [b.king[c] for c in range(4)]
            castles = [m for m in lm_o if m[0] in kings and max(abs(m[0] // R - m[1] // R), abs(m[0] % R - m[1] % R)) == 2]
            n_castle += len(castles)
            flats = sorted(set(x[2] for x in lm_o))
            pick = castles[0][2] if castles and rng.random() < 0.5 else flats[rng.randrange(len(flats))]
            b, rc = orc.take_action(b, R, pick)
            fb = eng.take_action([fb], [pick])[0]
            assert rc == 0 and fpc_ffi.lists_of(fb) == orc.lists_of(b)
        roots_o.append(orc.board_from_dict(R, turn, [list(e) for e in entries], castle=[3, 3, 3, 3]) if orc.game_result(orc.clone(b), R, INV) != 0 else b)
    assert n_castle > 0
    ev = evaluators.make("hash", R)
    rc, oref = orc.search([orc.clone(b) for b in roots_o], R, INV, sims, 3.0, ev)
    assert rc == 0
    roots = []
    for b in roots_o:
        fb = fpc_ffi.board_from_lists(R, b.turn, orc.lists_of(b))
        for c in range(4):
            fb.castle[c] = b.castle[c]
        roots.append(fb)
    res = run_external_search(eng, backend, roots, sims, 3.0, ev)
    _compare_search(res, oref, ("castling",))
    eng.close()
    return n_castle


def case_arena_vs_oracle(backend, R, n_pairs=3, sims=24, max_len=30, seed=9, kinds=("hash", "ramp")):
    """BASELINE configs[4] (arena eval, temperature 0): paired games between two evaluators through
    arena.play_paired, against the same loop run on the ORACLE: every ply's legal child set, visit
    counts and pick, every result and the adjudication must be identical."""
    import random
    import arena
    g = gold(R)
    INV = g["INV"]
    st = g["start"]
    rng = random.Random(seed)
    starts_o = []
    for _ in range(n_pairs):
        b = orc.board_from_dict(R, st["turn"], st["dict"])
        for _ply in range(rng.randrange(0, 24)):
            lm = orc.legal_moves(b, R, INV)
            flats = sorted(set(x[2] for x in lm))
            nb, rc = orc.take_action(b, R, flats[rng.randrange(len(flats))])
            assert rc == 0
            if orc.game_result(orc.clone(nb), R, INV) != 0:
                break
            b = nb
        starts_o.append(b)
    ev_a, ev_b = evaluators.make(kinds[0], R), evaluators.make(kinds[1], R)
    args = {"max_game_length": max_len}

    # ---- the oracle's arena (same batching, same rules)
    class OG:
        pass
    ogames = []
    for k, b in enumerate(starts_o):
        for a_team in (0, 1):
            o = OG(); o.state = orc.clone(b); o.a_team = a_team; o.plies = []; o.result = 0; o.winner = None
            ogames.append(o)
    live = list(ogames)
    for _ply in range(max_len):
        if not live:
            break
        for ev, batch in ((ev_a, [o for o in live if (o.state.turn & 1) == o.a_team]),
                          (ev_b, [o for o in live if (o.state.turn & 1) != o.a_team])):
            if not batch:
                continue
            boards = [o.state for o in batch]
            rc, res = orc.search(boards, R, INV, sims, 3.0, ev)
            assert rc == 0
            for o, r in zip(batch, res):
                flats = [c[0] for c in r["children"]]; visits = [c[1] for c in r["children"]]
                pick = arena.pick_argmax(flats, visits)
                o.plies.append((int(o.state.turn), flats, visits, pick))
                nb, rc2 = orc.take_action(r["board"], R, pick)      # the search leaves its list order in the root
                assert rc2 == 0
                o.state = nb
                gr = orc.game_result(o.state, R, INV)
                if gr != 0:
                    o.result = gr
                    o.winner = 0 if gr == 1 else (1 if gr == 2 else -1)
        live = [o for o in live if o.result == 0]

    # ---- the engine's arena
    eng = make_engine(backend, R, INV, max_games=2 * n_pairs, max_sims=sims)
    starts = [fpc_ffi.board_from_lists(R, b.turn, orc.lists_of(b)) for b in starts_o]
    games = arena.play_paired(lambda pods: run_external_search(eng, backend, pods, sims, 3.0, ev_a),
                              lambda pods: run_external_search(eng, backend, pods, sims, 3.0, ev_b), eng, starts, args)
    assert len(games) == len(ogames)
    n_plies = 0
    for ga, o in zip(games, ogames):
        assert len(ga.plies) == len(o.plies), (ga.gid, len(ga.plies), len(o.plies))
        for (t, fl, vi, pk), (ot, ofl, ovi, opk) in zip(ga.plies, o.plies):
            assert t == ot and [int(x) for x in fl] == ofl and [int(x) for x in vi] == ovi and pk == opk, (ga.gid, t)
        assert ga.result == o.result
        if o.result != 0:
            assert ga.winner == o.winner
        assert fpc_ffi.lists_of(ga.state) == orc.lists_of(o.state)
        n_plies += len(ga.plies)
    s = arena.summary(games)
    assert s["games"] == 2 * n_pairs and abs(s["score_a"] + s["score_b"] - s["games"]) < 1e-9
    eng.close()
    return n_plies


def case_fixed_rules_vs_oracle(backend, R, n_games=6, plies=60, sims=40, seed=31, rules=15, noise=True):
    """SURVEY 8f N4 -- NOT part of parity with the reference: the non-strict rule set (FPC_RULES_FIXED:
    AlphaZero PUCT with the child's value negated, per-sample rotation, un-shifted input planes, full
    moves = queen promotion / rook hop / castling rights) and root Dirichlet noise, engine against the
    oracle running the same rules: boards after every move, encodes, and searches bit for bit."""
    import random
    import positions
    INV = {8: 2, 10: 2, 13: 3, 14: 3}[R]
    turn, entries = positions.start_entries(R)
    castle = [3, 3, 3, 3] if R == 14 else None
    if R == 14:     # rooks, kings and pawns only, all rights set: castling becomes reachable
        entries = [e for e in entries if e[2] in (positions.PAWN, positions.ROOK, positions.KING)]
    rng = random.Random(seed)
    eng = make_engine(backend, R, INV, max_games=n_games, max_sims=sims)
    n_promo = n_castle = 0
    try:
        orc.set_rules(rules)
        eng.set_rules(rules)
        roots_o = []
        for _ in range(n_games):
            b = orc.board_from_dict(R, turn, [list(e) for e in entries], castle=castle)
            fb = fpc_ffi.board_from_dict(R, turn, entries, castle=castle, _lib=eng.L)
            for _ply in range(rng.randrange(plies // 2, plies)):
                if orc.game_result(b, R, INV) != 0:
                    break
                eng.game_result([fb])
                lm_o = orc.legal_moves(b, R, INV)
                lm_e = expand_promos(eng.legal_moves([fb])[0])
                assert lm_e == lm_o
                flats = sorted(set(x[2] for x in lm_o))
                kings = [b.king[c] for c in range(4)]
                castles = [m for m in lm_o if m[0] in kings and max(abs(m[0] // R - m[1] // R), abs(m[0] % R - m[1] % R)) == 2]
                pick = castles[0][2] if castles and rng.random() < 0.7 else flats[rng.randrange(len(flats))]
                n_castle += bool(castles and pick == castles[0][2])
                queens_before = sum(1 for c in range(4) for i in range(b.plen[c]) if ((b.sq[b.pl[c][i]] >> 2) & 7) == 4)
                b, rc = orc.take_action(b, R, pick)
                fb = eng.take_action([fb], [pick])[0]
                assert rc == 0
                n_promo += sum(1 for c in range(4) for i in range(b.plen[c]) if ((b.sq[b.pl[c][i]] >> 2) & 7) == 4) > queens_before
                assert bytes(fb.sq) == bytes(b.sq) and fpc_ffi.lists_of(fb) == orc.lists_of(b), (_ply, pick)
                assert [fb.castle[c] for c in range(4)] == [b.castle[c] for c in range(4)] and fb.turn == b.turn
                assert np.array_equal(eng.encode([fb]), orc.encode([b], R))
            if orc.game_result(orc.clone(b), R, INV) != 0:
                b = orc.board_from_dict(R, turn, [list(e) for e in entries], castle=castle)
            roots_o.append(b)
        ev = evaluators.make("hash", R)
        gamma = None
        if noise:
            gamma = np.random.default_rng(seed).standard_gamma(0.3, size=(n_games, fpc_ffi.MAX_MOVES)).astype(np.float32)
            orc.set_root_noise(gamma, 0.25)
            eng.set_root_noise(gamma, 0.25)
        rc, oref = orc.search([orc.clone(b) for b in roots_o], R, INV, sims, 3.0, ev)
        assert rc == 0
        roots = []
        for b in roots_o:
            fb = fpc_ffi.board_from_lists(R, b.turn, orc.lists_of(b))
            for c in range(4):
                fb.castle[c] = b.castle[c]
            roots.append(fb)
        res = run_external_search(eng, backend, roots, sims, 3.0, ev, fused=True)
        _compare_search(res, oref, ("fixed", R, rules))
        if noise:      # the noise really entered: strict-rule priors differ
            eng.set_root_noise(None, 0.0)
            res2 = run_external_search(eng, backend, [fpc_ffi.clone_board(r) for r in roots], sims, 3.0, ev)
            assert not np.array_equal(res["prior"], res2["prior"])
    finally:
        orc.set_rules(0)
        orc.set_root_noise(None)
        eng.close()
    return n_promo, n_castle


def load_attack_golden(R):
    import gzip
    import json
    import os
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_attack_r%d.json.gz" % R), "rt") as f:
        return json.load(f)


def case_attack_maps(backend, R, stride=1):
    """fpc_boards_attack_maps (k_attack_maps) against the golden dump of the reference's GetAttackedSquaresPlayers /
    GetAttackedSquaresTeams / IsAttackedByPlayer (oracle/gen_attack_golden.py) AND against the oracle's maps, bit for bit;
    the boards come back untouched (the queries are const in the reference)."""
    from oracle import orc
    g = load_attack_golden(R)
    cases = g["cases"][::stride]
    INV = {8: 2, 14: 3}[R]
    eng = make_engine(backend, R, INV, max_games=8, max_sims=4)
    boards = [fpc_ffi.board_from_lists(R, c["turn"], c["pl"]) for c in cases]
    before = [bytes(b) for b in boards]
    maps = eng.attack_maps(boards)
    assert [bytes(b) for b in boards] == before
    for c, m in zip(cases, maps):
        for colour in range(4):
            assert [int(x) for x in np.nonzero(m[colour])[0]] == c["players"].get(str(colour), []), (c["pos"], colour)
        for team in range(2):
            assert [int(x) for x in np.nonzero(m[4 + team])[0]] == c["teams"].get(str(team), []), (c["pos"], team)
        if "by_player" in c:
            assert m[:4].tolist() == c["by_player"], c["pos"]
        ob = orc.board_from_lists(R, c["turn"], c["pl"])
        assert np.array_equal(m, orc.attack_maps(ob, R, INV)), c["pos"]
    eng.close()
    return len(cases)
