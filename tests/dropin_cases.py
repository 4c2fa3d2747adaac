"""The drop-in Python surface (alphazero_cpp / mcts / four_player_chess_board / fen_parser in
alphazero-4-player-chess_amd/) exercised the way the reference's own training loop uses it
(alphazero.py:81-144, mcts.py:17-43), checked against the golden vectors of the real reference."""
import numpy as np
import torch

import fpc_ffi
from fpc_testlib import emul_lib, gold


def setup(backend, R):
    import alphazero_cpp as az
    az.configure(R)
    if backend == "emul":      # test-side injection of the wavefront-emulator build (never done by the product)
        eng = fpc_ffi.Engine(R, az.Board.invalidArea(), max_games=16, max_sims=128, _lib=emul_lib())
        eng.nn_dtype, eng.weights_version = 0, None
        az._engine, az._engine_cap = eng, (16, 128)
    return az


class Eval:
    """same synthetic evaluators as oracle/gen_golden.py (torch form)"""
    def __init__(self, kind, R, device="cpu"):
        self.kind, self.device, self.R = kind, device, R
        self.A = (8 * R + 8) * R * R
        self.widx = ((torch.arange(24 * R * R, dtype=torch.int64) * 2654435761) % (1 << 32)).view(1, 24, R, R)

    def __call__(self, x):
        x = x.cpu()
        B, A = x.shape[0], self.A
        if self.kind == "zero":
            return torch.zeros(B, A), torch.zeros(B, 1)
        h = ((x.to(torch.int64) * self.widx).sum(dim=(1, 2, 3))) % (1 << 32)
        i = torch.arange(A, dtype=torch.int64).view(1, A)
        u = ((h.view(B, 1) * 2246822519 + i * 40503 + ((i * i) % 8191) * 69069) % (1 << 32)) >> 16
        logits = u.to(torch.float32) / 8192.0 - 4.0
        v = ((h % 9).to(torch.float32) - 4) / 4
        return logits, v.view(B, 1)


def case_reference_style_usage(backend, R):
    az = setup(backend, R)
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    from mcts import MCTS
    g = gold(R)
    assert FourPlayerChess.nRows() == R and FourPlayerChess.action_space_size == g["A"]
    assert FourPlayerChess.num_action_channels == g["A_ch"] and FourPlayerChess.state_space_size == g["state_space_size"]
    assert az.Move.num_queen_moves == g["num_queen_moves"]
    board_args = parse_board_args_from_fen(FourPlayerChess.start_fen, FourPlayerChess.nCols())
    game = FourPlayerChess(*board_args)
    pl = [[[pp.GetLocation().GetRow() * R + pp.GetLocation().GetCol(), int(pp.GetPiece().GetPieceType())] for pp in col]
          for col in game.GetPieces()]
    assert pl == g["start"]["after_ctor"]["pl"]
    rec = g["playouts"][0][0]
    assert int(game.GetGameResult()) == rec["result"]
    lm = game.GetLegalMoves()
    got = [[m.From().GetRow() * R + m.From().GetCol(), m.To().GetRow() * R + m.To().GetCol(), m.GetFlatIndex()] for m in lm]
    assert got == rec["legal"]
    enc = az.Board.GetEncodedState(game, "cpu")
    assert tuple(enc.shape) == (1, 24, R, R)
    assert torch.nonzero(enc.flatten()).flatten().tolist() == rec["enc"]
    mask = FourPlayerChess.get_legal_moves_mask([game], "cpu")
    assert sorted(set(x[2] for x in rec["legal"])) == torch.nonzero(mask.flatten()).flatten().tolist()
    assert "Turn: Player(RED)" in str(game)
    assert str(game.GetPieces()[0][0]).startswith("Red ")
    # a few plies of the reference's play() loop on the recorded picks
    state = FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, R))
    for rec in g["playouts"][0][:6]:
        assert int(state.GetGameResult()) == rec["result"]
        state.GetLegalMoves()
        nxt = state.TakeAction(az.Move(rec["pick"]))
        nxt.SetRootState(state.GetRootState())
        state = nxt
    # MCTS.search with synthetic evaluators, reference-style read-out (alphazero.py:104-110)
    for rec in g["searches"]:
        if rec["kind"] not in ("zero", "hash") or rec["sims"] != 100 or len(rec["before"]) != 1:
            continue
        game = FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, R))
        mcts = MCTS(FourPlayerChess, Eval(rec["kind"], R), {"C": rec["C"], "num_searches": rec["sims"], "pool_size": 10})
        roots = mcts.search([game])
        action_probs = torch.zeros(FourPlayerChess.action_space_size)
        for child in roots[0].GetChildren():
            action_probs[child.GetMoveMade().GetFlatIndex()] = child.GetVisitCount()
        ref = rec["roots"][0]
        assert roots[0].GetVisitCount() == ref["root_n"]
        assert [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in roots[0].GetChildren()] == [[c[0], c[1]] for c in ref["children"]]
        assert [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in roots[0].GetChildren()[0].GetChildren()] == ref["children"][0][2]
        assert game.GetRootNode() is roots[0]
        assert float(action_probs.sum()) == sum(c[1] for c in ref["children"])
        state_pl = [[[pp.GetLocation().GetRow() * R + pp.GetLocation().GetCol(), int(pp.GetPiece().GetPieceType())]
                     for pp in col] for col in game.GetPieces()]
        assert state_pl == ref["after"]
        game.AppendToMemory(az.MemoryEntry(game, action_probs / action_probs.sum()))
        assert len(game.GetMemory()) == 1
    return True


def case_training_loop(backend, R=8):
    """AlphaZero.learn(): self-play on the engine -> replay buffer -> PyTorch optimiser step ->
    weights pushed back into the engine for the next search (N2/N3 of SURVEY 8f)."""
    az = setup(backend, R)
    from alphazero import AlphaZero
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    import net
    torch.manual_seed(0)
    model = net.ResNet(FourPlayerChess, 1, 64, "cpu")
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    args = {"C": 3, "num_searches": 8, "num_iterations": 1, "num_games": 4, "num_parallel_games": 4, "batch_size": 8,
            "temperature": 1.1, "heuristic_weight": 0.02, "max_game_length": 6, "replay_buffer_capacity": 150,
            "validation_buffer_capacity": 30, "pool_size": 10, "nn_dtype": 1}
    evaluator = None
    if backend == "emul":      # the emulator build has no MFMA network: plug the model in through the evaluator seam
        class Wrap:
            device = "cpu"

            def __call__(self, x):
                with torch.no_grad():
                    return model(x)
        evaluator = Wrap()
    init = parse_board_args_from_fen(FourPlayerChess.start_fen, R)
    a = AlphaZero(model, optimizer, FourPlayerChess, args, init, evaluator=evaluator, seed=3)
    before = [p.detach().clone() for p in model.parameters()]
    a.learn()
    assert len(a.experience_buffer) + len(a.validation_buffer) == 2 * 4 * 6
    assert a.history and all(np.isfinite(h["loss"]) for h in a.history)
    assert any(not torch.equal(b, p.detach()) for b, p in zip(before, model.parameters()))
    return len(a.history)


# ---- host forms of the Node API (wrapper.cpp:233-253): the reference's per-simulation loop runs on the shim ----
def _host_tree_search(az, game_type, games, evaluator, C, sims):
    """A per-simulation search written against nothing but the bound API of the reference (Node.ChooseLeaf,
    Board.GetEncodedStates / ParseActionspace, get_legal_moves_mask, Node.BackpropagateNodes / ExpandNodes) --
    the contract its own mcts.py:17-89 programs against, in our own words."""
    roots = []
    for g in games:
        roots.append(az.Node(C, g, visit_count=1))
        g.SetRootNode(roots[-1])
    live = list(roots)
    pool = az.BoardPool(10)
    for _sim in range(sims):
        leaves, still = [], []
        for r in live:                      # a root whose descent ended in a terminal leaf leaves the search (Q5)
            leaf = r.ChooseLeaf()
            if leaf is not None:
                leaves.append(leaf)
                still.append(r)
        live = still
        if not leaves:
            continue
        states = [l.GetState() for l in leaves]
        logits, value = evaluator(game_type.GetEncodedStates(states, "cpu"))
        p = game_type.ParseActionspace(torch.softmax(logits, dim=1), states[0].GetTurn())
        p = p * game_type.get_legal_moves_mask(states, "cpu")
        p = p / p.sum(dim=(1, 2, 3), keepdim=True)
        az.Node.BackpropagateNodes(leaves, value.squeeze(1))
        nz = torch.nonzero(p)
        az.Node.ExpandNodes(leaves, p, nz.tolist(), p[tuple(nz.t())].tolist(), pool)
    return roots


def _load_reference_mcts():
    """the reference's OWN src/py/mcts.py (only where /root/reference exists: the build container), bound to our
    shim: `alphazero_cpp` resolves to alphazero-4-player-chess_amd/alphazero_cpp.py; its one foreign import
    (line_profiler_pycharm.profile, a PyCharm plugin) gets an identity decorator"""
    import importlib.util
    import os
    import sys
    import types
    path = "/root/reference/src/py/mcts.py"
    if not os.path.exists(path):
        return None
    if "line_profiler_pycharm" not in sys.modules:
        stub = types.ModuleType("line_profiler_pycharm")
        stub.profile = lambda f: f
        sys.modules["line_profiler_pycharm"] = stub
    spec = importlib.util.spec_from_file_location("reference_mcts_py", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def case_host_tree(backend, R, max_cases=3, max_sims=100, use_reference_mcts=False):
    """Golden MCTS.search cases of the real reference, re-run through the HOST forms of Node (tree bookkeeping
    in the shim, board operations through the engine's C-ABI): root and second-level visit counts must match
    the reference bit for bit; with use_reference_mcts the loop that runs IS the reference's own mcts.py."""
    az = setup(backend, R)
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    ref_mod = _load_reference_mcts() if use_reference_mcts else None
    if use_reference_mcts and ref_mod is None:
        return -1
    g = gold(R)
    done = 0
    for rec in g["searches"]:
        if rec["kind"] not in ("zero", "hash") or rec["sims"] > max_sims or len(rec["before"]) != 1 or done >= max_cases:
            continue
        game = FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, R))
        start_pl = [[[pp.GetLocation().GetRow() * R + pp.GetLocation().GetCol(), int(pp.GetPiece().GetPieceType())]
                     for pp in col] for col in game.GetPieces()]
        if start_pl != rec["before"][0]["pl"] or int(game.GetTurn().GetColor()) != rec["before"][0]["turn"]:
            continue                        # a case that does not start from the start position
        ev = Eval(rec["kind"], R)
        if ref_mod is not None:
            m = ref_mod.MCTS(FourPlayerChess, ev, {"C": rec["C"], "num_searches": rec["sims"], "pool_size": 10})
            roots = m.search([game])
        else:
            roots = _host_tree_search(az, FourPlayerChess, [game], ev, rec["C"], rec["sims"])
        ref = rec["roots"][0]
        assert roots[0].GetVisitCount() == ref["root_n"]
        assert [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in roots[0].GetChildren()] == [[c[0], c[1]] for c in ref["children"]]
        assert [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in roots[0].GetChildren()[0].GetChildren()] == ref["children"][0][2]
        state_pl = [[[pp.GetLocation().GetRow() * R + pp.GetLocation().GetCol(), int(pp.GetPiece().GetPieceType())]
                     for pp in col] for col in game.GetPieces()]
        assert state_pl == ref["after"]
        done += 1
    return done


def case_play_loop_prefetch(backend, R, games=6, plies=4, sims=24):
    """alphazero.py:99-144 written the way the reference writes it -- per game: GetChildren / GetMoveMade().GetFlatIndex()
    / GetVisitCount, MemoryEntry, TakeAction(Move(flat)), SetRootState, GetGameResult -- over MCTS.search.  The shim
    answers TakeAction / GetGameResult of a searched root from ONE batched prefetch per search
    (alphazero_cpp._SearchBatch); every successor position (bytes, i.e. incl. piece-list order, before and after
    GetGameResult) and every result must equal what the engine returns for the same calls made one by one."""
    az = setup(backend, R)
    import ctypes as C
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    from mcts import MCTS
    mcts = MCTS(FourPlayerChess, Eval("hash", R), {"C": 3.0, "num_searches": sims, "pool_size": 10})
    states = [FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, R)) for _ in range(games)]
    rng = np.random.default_rng(7)
    checked = 0
    for ply in range(plies):
        roots = mcts.search(states)
        eng = az.engine()
        for i in reversed(range(len(states))):
            state = states[i]
            probs = np.zeros(FourPlayerChess.action_space_size)
            kids = roots[i].GetChildren()
            assert kids and all(isinstance(c, az.Node) and isinstance(c.GetMoveMade(), az.Move) for c in kids)
            for child in kids:
                probs[child.GetMoveMade().GetFlatIndex()] = child.GetVisitCount()
            f, v = roots[i].child_arrays()
            assert [int(x) for x in f] == [c.GetMoveMade().GetFlatIndex() for c in kids] and [int(x) for x in v] == [c.GetVisitCount() for c in kids]
            assert kids[0].GetMoveMade().From().Present() and az.Move(kids[0].GetMoveMade()).GetFlatIndex() == kids[0].GetMoveMade().GetFlatIndex()
            state.AppendToMemory(az.MemoryEntry(state, probs / probs.sum()))
            pick = int(rng.choice(np.nonzero(probs)[0]))
            # the same two calls made directly on the engine, one board at a time
            want_pre = eng.take_action([fpc_ffi.clone_board(state._b)], [pick])[0]
            want_post = fpc_ffi.clone_board(want_pre)
            want_res = eng.game_result([want_post])[0]
            nxt = state.TakeAction(az.Move(pick))
            assert nxt._gr_cache is not None, "a root child's successor must come from the prefetch"
            assert bytes(nxt._b) == bytes(want_pre)
            nxt.SetRootState(state.GetRootState())
            res = nxt.GetGameResult()
            assert int(res) == want_res and bytes(nxt._b) == bytes(want_post)
            assert nxt._gr_cache is None and int(nxt.GetGameResult()) == eng.game_result([fpc_ffi.clone_board(want_post)])[0]
            checked += 1
            if res != az.GameResult.IN_PROGRESS:
                del states[i]
            else:
                states[i] = nxt
        if not states:
            break
    # a position that is not the searched root falls back to the direct path
    if states:
        roots = mcts.search(states[:1])
        moved = states[0].TakeAction(az.Move(int(roots[0].child_arrays()[0][0])))
        moved.SetRootNode(roots[0])                      # wrong on purpose: not the root's position
        lm = moved.GetLegalMoves()
        other = moved.TakeAction(lm[0])
        assert other._gr_cache is None
    return checked


def case_attacked_square_methods(backend, R, max_cases=24):
    """Board.GetAttackedSquaresPlayers / GetAttackedSquaresTeams / IsAttackedByPlayer / GetSimpleState through the
    drop-in surface (wrapper.cpp:201-206), the way the reference's reviewer calls them, against the golden dump of the
    real reference (tests/golden/ref_attack_r*.json.gz): dict keys, locations and their order."""
    from engine_cases import load_attack_golden
    az = setup(backend, R)
    from four_player_chess_board import FourPlayerChess
    g = load_attack_golden(R)
    done = 0
    for c in [c for c in g["cases"] if "by_player" in c][:max_cases]:
        l2p = {}
        for colour, col in enumerate(c["pl"]):
            for sq, typ in col:
                l2p[az.BoardLocation(sq // R, sq % R)] = az.Piece(az.PlayerColor(colour), az.PieceType(typ))
        b = FourPlayerChess(az.Player(az.PlayerColor(c["turn"])), l2p)
        sqs = lambda locs: [int(l.GetRow()) * R + int(l.GetCol()) for l in locs]
        players = b.GetAttackedSquaresPlayers()
        assert {str(int(k)): sqs(v) for k, v in players.items()} == c["players"]
        assert all(isinstance(k, az.PlayerColor) for k in players)
        teams = b.GetAttackedSquaresTeams()
        assert {str(int(k)): sqs(v) for k, v in teams.items()} == c["teams"]
        for colour in range(4):
            for sq in range(0, R * R, 7):
                assert b.IsAttackedByPlayer(az.BoardLocation(sq // R, sq % R), az.PlayerColor(colour)) == bool(c["by_player"][colour][sq])
        assert b.IsAttackedByPlayer(az.BoardLocation(), az.PlayerColor.RED) is False
        st = b.GetSimpleState()
        assert int(st.turn.GetColor()) == c["simple"]["turn"]
        assert {str(int(k)): sqs(v) for k, v in st.attackedSquares.items()} == c["simple"]["attacked"]
        got = [[[int(pp.GetLocation().GetRow()) * R + int(pp.GetLocation().GetCol()), int(pp.GetPiece().GetPieceType())] for pp in col] for col in st.pieces]
        assert got == c["simple"]["pieces"]                 # through the constructor: the reference's own list order
        assert len(st.castlingRights) == 4 and not any(cr.Kingside() or cr.Queenside() for cr in st.castlingRights)
        done += 1
    return done
