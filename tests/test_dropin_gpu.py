import pytest

import dropin_cases as dc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R", [8, 14])
def test_reference_style_usage(R):
    assert dc.case_reference_style_usage("gpu", R)


@pytest.mark.parametrize("R", [8, 14])
def test_play_loop_takes_successors_from_one_batched_prefetch(R):
    """the reference's per-game play loop over MCTS.search: TakeAction / GetGameResult of the searched roots come from
    one batched prefetch per search and equal the one-by-one engine calls byte for byte (incl. piece-list order)"""
    assert dc.case_play_loop_prefetch("gpu", R, games=24, plies=6, sims=40) >= 24


def test_native_resnet_search_through_mcts():
    """MCTS(gameType, ResNet, args).search: weights exported + fused on-device search."""
    import torch
    az = dc.setup("gpu", 8)
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    from mcts import MCTS
    import net
    torch.manual_seed(0)
    model = net.ResNet(FourPlayerChess, 2, 64, "cpu").eval()
    games = [FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, 8)) for _ in range(6)]
    mcts = MCTS(FourPlayerChess, model, {"C": 3, "num_searches": 50, "pool_size": 10, "nn_dtype": 1})
    roots = mcts.search(games)
    for r in roots:
        assert r.GetVisitCount() == 51
        assert sum(c.GetVisitCount() for c in r.GetChildren()) == len(r.GetChildren()) + 50 - 1    # quirk Q1
    first = [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in roots[0].GetChildren()]
    assert all([[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in r.GetChildren()] == first for r in roots)


def test_training_loop_native_network():
    assert dc.case_training_loop("gpu") >= 1


def test_native_search_policy_head_option():
    """args["policy_head"] = "legal" (opt-in legal-moves-only policy head) through the drop-in MCTS:
    same children, same visit counts as the default full head on the same network and positions."""
    import torch
    az = dc.setup("gpu", 8)
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    from mcts import MCTS
    import net
    torch.manual_seed(1)
    model = net.ResNet(FourPlayerChess, 2, 64, "cpu").eval()
    out = {}
    for head in ("full", "legal"):
        games = [FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, 8)) for _ in range(5)]
        mcts = MCTS(FourPlayerChess, model, {"C": 3, "num_searches": 40, "pool_size": 10, "nn_dtype": 1, "policy_head": head})
        roots = mcts.search(games)
        out[head] = [[[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount()] for c in r.GetChildren()] for r in roots]
    assert out["full"] == out["legal"]


def test_native_tuples_and_rccl_single_rank():
    """fpc_collect_tuples / fpc_tuples_set_z on the device, then the RCCL path of the C-ABI with a
    one-rank communicator (librccl bound at run time, ncclCommInitRank + two ncclAllGather from the C++
    host): the gathered PODs equal the collected ones; dense (state, pi, z) rebuilds from them."""
    import numpy as np
    import evaluators
    import fpc_ffi
    import positions
    import tuples
    from fpc_testlib import make_engine, run_external_search
    R, G, sims = 8, 5, 24
    eng = make_engine("gpu", R, 2, max_games=G, max_sims=sims)
    turn, entries = positions.start_entries(R)
    boards = [fpc_ffi.board_from_dict(R, turn, entries) for _ in range(G)]
    ev = evaluators.make("hash", R)
    eng.tuples_reserve(3 * G)
    ids = [10 + 3 * g for g in range(G)]
    results = []
    for ply in range(3):
        res = run_external_search(eng, "gpu", boards, sims, 3.0, ev)
        eng.collect_tuples(ids, ply)
        results.append((res, [fpc_ffi.clone_board(b) for b in boards]))
        boards = eng.take_action(boards, [int(res["flat"][g, 0]) for g in range(G)])
    eng.tuples_set_z(ids[:3], [1.0, -1.0, 0.5], [-1.0, 1.0, -0.5])
    arr, n = eng.tuples_read()
    recs = tuples.records_of(arr, n, R)
    assert n == 3 * G
    for ply, (res, roots) in enumerate(results):
        for g in range(G):
            r = recs[ply * G + g]
            k = int(res["n_children"][g])
            assert r["game"] == ids[g] and r["ply"] == ply and r["turn"] == roots[g].turn
            assert r["mailbox"].tobytes() == bytes(roots[g].sq)[:R * R]
            assert r["flat"].tolist() == res["flat"][g, :k].tolist() and r["visits"].tolist() == res["visits"][g, :k].tolist()
            want_z = {0: (1.0, -1.0), 1: (-1.0, 1.0), 2: (0.5, -0.5)}.get(g, (0.0, 0.0))[roots[g].turn & 1]
            assert r["z"] == want_z
    eng.comm_init(fpc_ffi.comm_unique_id(), 0, 1)
    counts, garr, total = eng.allgather_tuples()
    assert total == n and counts[0] == n
    assert bytes(memoryview(garr).cast("B")[:n * 1280]) == bytes(memoryview(arr).cast("B")[:n * 1280])
    enc, pi, z = tuples.dense_batch(eng, tuples.records_of(garr, total, R))
    assert tuple(enc.shape) == (n, 24, R, R) and abs(float(pi.sum(dim=1).min()) - 1.0) < 1e-6 and float(z.abs().max()) == 1.0
    for ply, (res, roots) in enumerate(results):
        assert np.array_equal(enc[ply * G].numpy(), eng.encode([roots[0]])[0])
    eng.close()


@pytest.mark.parametrize("R", [8, 14])
def test_host_node_forms_reproduce_reference_searches(R):
    """Node.ChooseLeaf / SelectChild / Backpropagate(Nodes) / ExpandNodes as host methods (wrapper.cpp:233-253);
    board operations on the GPU through the C-ABI; golden visit counts of the real reference."""
    assert dc.case_host_tree("gpu", R, max_cases=3, max_sims=100) >= 1


@pytest.mark.parametrize("R", [8, 14])
def test_attacked_square_methods_of_the_binding_surface(R):
    assert dc.case_attacked_square_methods("gpu", R, max_cases=24) >= 24
