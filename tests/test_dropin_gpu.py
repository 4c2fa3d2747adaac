import pytest

import dropin_cases as dc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R", [8, 14])
def test_reference_style_usage(R):
    assert dc.case_reference_style_usage("gpu", R)


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
