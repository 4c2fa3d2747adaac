"""CPU coverage of the product's tree-kernel source + host engine, executed on the 64-lane
wavefront emulator (tests/emul/).  Same cases as tests/test_engine_gpu.py, smaller sizes."""
import pytest

import engine_cases as ec


@pytest.mark.parametrize("R", [8, 14])
def test_static(R):
    ec.case_static("emul", R)


@pytest.mark.parametrize("R,games,plies", [(8, 6, 60), (14, 3, 24)])
def test_playouts(R, games, plies):
    n_pos, n_term, n_child = ec.case_playouts("emul", R, games, plies)
    assert n_pos > 50


@pytest.mark.parametrize("R", [8, 14])
def test_batch_encode(R):
    ec.case_batch_encode("emul", R)


def test_search_golden_r8():
    assert ec.case_search_golden("emul", 8, max_sims=100) >= 6


def test_search_golden_r14():
    assert ec.case_search_golden("emul", 14, max_sims=100, max_cases=3) >= 3


def test_search_random_vs_oracle():
    ec.case_search_random_vs_oracle("emul", 8, n_games=5, sims=40, seed=11)


def test_selfplay_trace_r8():
    assert ec.case_selfplay_trace("emul", 8, max_traces=1) > 100     # the GPU suite replays every trace at both sizes


@pytest.mark.parametrize("R,INV", [(10, 2), (13, 3), (9, 2), (11, 3), (12, 3)])
def test_other_board_sizes_vs_oracle(R, INV):
    assert ec.case_other_sizes_vs_oracle("emul", R, INV, n_games=3, sims=20)


def test_castling_vs_oracle():
    assert ec.case_castling_vs_oracle("emul", n_games=2, plies=30, sims=16) > 0


def test_arena_vs_oracle():
    """configs[4]: paired temperature-0 arena games, engine vs oracle, every ply bit-exact"""
    assert ec.case_arena_vs_oracle("emul", 8, n_pairs=2, sims=16, max_len=14) > 20


@pytest.mark.parametrize("R", [8, 14])
def test_fixed_rules_and_root_noise_vs_oracle(R):
    """N4: non-strict rule set + root Dirichlet noise, engine vs oracle under the same rules"""
    n_promo, n_castle = ec.case_fixed_rules_vs_oracle("emul", R, n_games=3, plies=70 if R == 8 else 30, sims=16)
    assert (n_promo > 0) if R == 8 else (n_castle > 0)


def test_step_entry_points_refuse_bad_calls():
    """Error behaviour of the step-wise search entry points (fpc_search_select / expand / expand_select):
    no search begun -> ESTATE; null logits -> EINVAL; more simulations than max_sims -> ECAPACITY."""
    import numpy as np
    import evaluators
    import fpc_ffi
    from fpc_testlib import gold, make_engine
    R = 8
    g = gold(R)
    st = g["start"]
    eng = make_engine("emul", R, g["INV"], max_games=2, max_sims=3)
    try:
        with pytest.raises(RuntimeError, match="fpc_search_begin"):
            eng.search_select()
        with pytest.raises(RuntimeError, match="fpc_search_begin"):
            eng.search_expand_select(0, 0)
        roots = [fpc_ffi.board_from_dict(R, st["turn"], [tuple(e) for e in st["dict"]], _lib=eng.L) for _ in range(2)]
        eng.search_begin(roots, 3.0)
        n_live, enc_ptr = eng.search_select()
        assert n_live == 2
        with pytest.raises(RuntimeError, match="null"):
            eng.search_expand_select(0, 0)
        ev = evaluators.make("hash", R)
        import ctypes as C
        for _ in range(2):      # simulations 2 and 3 through the fused entry point
            enc = np.ctypeslib.as_array(C.cast(enc_ptr, C.POINTER(C.c_float)), shape=(2, 24, R, R))
            lg, v = ev(enc)
            lg = np.ascontiguousarray(lg, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
            n_live, enc_ptr = eng.search_expand_select(lg.ctypes.data, v.ctypes.data)
        enc = np.ctypeslib.as_array(C.cast(enc_ptr, C.POINTER(C.c_float)), shape=(2, 24, R, R))
        lg, v = ev(enc)
        lg = np.ascontiguousarray(lg, dtype=np.float32); v = np.ascontiguousarray(v, dtype=np.float32)
        with pytest.raises(RuntimeError, match="max_sims"):        # a 4th selection would exceed max_sims = 3
            eng.search_expand_select(lg.ctypes.data, v.ctypes.data)
        eng.search_expand(lg.ctypes.data, v.ctypes.data)            # the plain expansion of the last one is fine
        res = eng.search_results(roots=roots)
        assert list(res["sims_done"]) == [3, 3] and list(res["root_n"]) == [4, 4]
    finally:
        eng.close()


@pytest.mark.parametrize("R", [8, 14])
def test_attack_maps_vs_reference_and_oracle(R):
    assert ec.case_attack_maps("emul", R, stride=4) >= 80
