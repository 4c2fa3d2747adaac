"""Cases built on the fixtures written by oracle/gen_net_golden.py (tests/golden/net_*.npz,
recnet_*.npz, ref_extra_*.json.gz): everything recorded from the reference's OWN net.py / mcts.py /
fen_parser.py / start_fens.py beyond the first golden file.  Shared by the CPU tests (oracle +
wavefront emulator) and the GPU tests (HIP engine through the C-ABI)."""
import gzip
import json
import os

import numpy as np

import fpc_ffi
from fpc_testlib import GOLD, gold, make_engine, run_external_search
from oracle import orc

INV_OF = {8: 2, 10: 2, 13: 3, 14: 3}


class Spec:
    """the gameType attributes net.ResNet reads (wrapper.cpp:176-181, :209-210)"""
    def __init__(self, R):
        self.R = R
        self.num_state_channels = 24
        self.num_action_channels = 8 * R + 8
        self.action_space_size = self.num_action_channels * R * R
        self.state_space_size = 24 * R * R

    def nRows(self):
        return self.R

    def nCols(self):
        return self.R


def load_net_fixture(R, blocks, hidden):
    z = np.load(os.path.join(GOLD, "net_r%d_b%d_h%d.npz" % (R, blocks, hidden)))
    fx = {k: z[k] for k in z.files}
    fx["meta"] = json.loads(str(fx["meta"]))
    fx["pnames"] = json.loads(str(fx["pnames"]))
    return fx


def load_extra(R):
    with gzip.open(os.path.join(GOLD, "ref_extra_r%d.json.gz" % R), "rt") as f:
        return json.load(f)


def perturb_bn(model, seed):
    """same rule as oracle/gen_net_golden.py"""
    import torch
    g = torch.Generator().manual_seed(seed + 1)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)


def check_param_sums(model, names, sums):
    """the module holds the weights the reference's net.py produced under the same seed"""
    sd = model.state_dict()
    got = [k for k, v in sd.items() if v.dtype.is_floating_point]
    assert got == list(names), "state_dict keys differ from the reference's net.py"
    for k, (s, a) in zip(names, np.asarray(sums)):
        d = sd[k].detach().double()
        assert abs(float(d.sum()) - s) <= 1e-9 * max(1.0, abs(a)) and abs(float(d.abs().sum()) - a) <= 1e-9 * max(1.0, abs(a)), k


def fixture_model(fx):
    """our net.py module, same seed + BN rule as the fixture, proven equal to the reference's by checksum"""
    import torch
    import net
    m = fx["meta"]
    torch.manual_seed(m["seed"])
    model = net.ResNet(Spec(m["R"]), m["blocks"], m["hidden"], "cpu")
    perturb_bn(model, m["seed"])
    model.eval()
    check_param_sums(model, fx["pnames"], fx["psums"])
    return model


def fixture_boards(fx):
    g = gold(fx["meta"]["R"])
    R = fx["meta"]["R"]
    out = []
    for gi, pi in fx["pos"]:
        snap = g["playouts"][int(gi)][int(pi)]["before"]
        out.append(fpc_ffi.board_from_lists(R, snap["turn"], snap["pl"]))
    return out


# ---- recorded-network search -------------------------------------------------
def load_recnet(R):
    z = np.load(os.path.join(GOLD, "recnet_r%d.npz" % R))
    rec = {k: z[k] for k in z.files}
    rec["meta"] = json.loads(str(rec["meta"]))
    return rec


class Replay:
    """evaluator that replays the reference network's recorded outputs call by call and checks that
    it is being shown exactly the inputs the reference's network saw (live leaves in game order;
    finished games are all-zero slots on our side and absent on the reference's)"""
    def __init__(self, rec, R):
        self.rec, self.R, self.call, self.row = rec, R, 0, 0
        self.A = (8 * R + 8) * R * R

    def __call__(self, enc):
        B = enc.shape[0]
        lg = np.zeros((B, self.A), np.float32)
        va = np.zeros(B, np.float32)
        live = [b for b in range(B) if enc[b].any()]
        assert self.call < len(self.rec["count"]), "more evaluator calls than the reference made"
        assert len(live) == int(self.rec["count"][self.call]), (self.call, len(live))
        for b in live:
            lo, hi = int(self.rec["enc_off"][self.row]), int(self.rec["enc_off"][self.row + 1])
            assert np.nonzero(enc[b].reshape(-1))[0].tolist() == self.rec["enc_idx"][lo:hi].tolist(), (self.call, b)
            lg[b] = self.rec["logits"][self.row]
            va[b] = self.rec["value"][self.row]
            self.row += 1
        self.call += 1
        return lg, va


def _check_roots(res, ref_roots, roots, eng=None):
    for gi, ref in enumerate(ref_roots):
        n = int(res["n_children"][gi])
        got = [[int(res["flat"][gi, k]), int(res["visits"][gi, k])] for k in range(n)]
        assert int(res["root_n"][gi]) == ref["root_n"]
        assert got == [[c[0], c[1]] for c in ref["children"]], gi
        assert fpc_ffi.lists_of(roots[gi]) == ref["after"], gi
        if eng is not None:
            for k, c in enumerate(ref["children"]):
                if k % 3 == 0:
                    assert eng.grandchildren(gi, k) == c[2], (gi, k)


def case_recorded_net_search(backend, R):
    """MCTS.search driven by the reference's ResNet, its outputs replayed: same visit counts."""
    rec = load_recnet(R)
    m = rec["meta"]
    if backend == "oracle":
        boards = [orc.board_from_lists(R, s["turn"], s["pl"]) for s in m["before"]]
        rp = Replay(rec, R)
        # the oracle shows the evaluator only the live leaves, like the reference
        rc, res = orc.search(boards, R, INV_OF[R], m["sims"], m["C"], rp)
        assert rc == 0 and rp.call == len(rec["count"])
        for o, ref in zip(res, m["roots"]):
            assert o["root_n"] == ref["root_n"] and o["children"] == [[c[0], c[1]] for c in ref["children"]]
            assert orc.lists_of(o["board"]) == ref["after"]
        return rp.call
    roots = [fpc_ffi.board_from_lists(R, s["turn"], s["pl"]) for s in m["before"]]
    eng = make_engine(backend, R, INV_OF[R], max_games=len(roots), max_sims=m["sims"])
    rp = Replay(rec, R)
    res = run_external_search(eng, backend, roots, m["sims"], m["C"], rp)
    assert rp.call == len(rec["count"])
    _check_roots(res, m["roots"], roots, eng)
    eng.close()
    return rp.call


def case_search_800(backend, R, kinds=None):
    """800 simulations per move (BASELINE configs[3]) against the reference's visit counts."""
    import evaluators
    x = load_extra(R)
    done = 0
    for rec in x["searches"]:
        if kinds and rec["kind"] not in kinds:
            continue
        ev = evaluators.make(rec["kind"], R)
        if backend == "oracle":
            boards = [orc.board_from_lists(R, s["turn"], s["pl"]) for s in rec["before"]]
            rc, res = orc.search(boards, R, INV_OF[R], rec["sims"], rec["C"], rec["kind"] if rec["kind"] in ("zero", "ramp") else ev)
            assert rc == 0
            for o, ref in zip(res, rec["roots"]):
                assert o["root_n"] == ref["root_n"] and o["children"] == [[c[0], c[1]] for c in ref["children"]]
                assert orc.lists_of(o["board"]) == ref["after"]
        else:
            roots = [fpc_ffi.board_from_lists(R, s["turn"], s["pl"]) for s in rec["before"]]
            eng = make_engine(backend, R, INV_OF[R], max_games=len(roots), max_sims=rec["sims"])
            res = run_external_search(eng, backend, roots, rec["sims"], rec["C"], ev)
            _check_roots(res, rec["roots"], roots, eng)
            eng.close()
        done += 1
    return done


def case_start_layouts():
    """all five start layouts (start_fens.py:1-67) as the reference's fen_parser reads them"""
    import positions
    x = load_extra(14)
    for name, ref in x["start_layouts"].items():
        turn, pieces, _k, _q = positions.parse_fen(getattr(positions, name), ref["size"])
        assert turn == ref["turn"], name
        assert [[r * ref["size"] + c, col, typ] for r, c, col, typ in pieces] == ref["dict"], name
    return len(x["start_layouts"])


def case_train_batch(backend):
    """N2: the (encoded state, dense pi, z) batch the trainer builds from self-play tuples and
    loss = cross_entropy + mse (alphazero.py:53-78, :181-209) against the reference-net fixture."""
    import torch
    import torch.nn.functional as F
    import net
    import tuples as tuples_mod
    R = 8
    x = load_extra(R)["train_batch"]
    A = (8 * R + 8) * R * R
    eng = make_engine(backend, R, INV_OF[R], max_games=4, max_sims=4)
    encs, pis, zs = [], [], []
    for t in x["tuples"]:
        b = fpc_ffi.board_from_lists(R, t["state"]["turn"], t["state"]["pl"])
        e = eng.encode([b])                                   # per tuple, own rotation (alphazero.py:71-73)
        assert np.nonzero(e.reshape(-1))[0].tolist() == t["enc"]
        encs.append(e)
        rec = {"flat": np.asarray([c[0] for c in t["pi"]], np.int64), "visits": np.asarray([c[1] for c in t["pi"]], np.int64)}
        # the compact record survives the wire format of the episode-end all-gather
        wire = tuples_mod.pack_record(R, bytes(b)[:R * R], b.turn, t["z"], rec["flat"], rec["visits"])
        back = tuples_mod.unpack_records(R, wire)[0]
        assert back["turn"] == b.turn and np.float32(back["z"]) == np.float32(t["z"])
        pis.append(tuples_mod.dense_pi(back, A))
        zs.append(t["z"])
    eng.close()
    enc = torch.from_numpy(np.concatenate(encs))
    pol = torch.stack(pis)
    zt = torch.tensor(zs, dtype=torch.float32).view(-1, 1)
    torch.manual_seed(x["seed"])
    model = net.ResNet(Spec(R), x["blocks"], x["hidden"], "cpu")
    check_param_sums(model, x["pnames"], x["psums"])
    for mode in ("train", "eval"):
        model.train(mode == "train")
        with torch.no_grad():
            out_policy, out_value = model(enc)
            pl = float(F.cross_entropy(out_policy, pol))
            vl = float(F.mse_loss(out_value.squeeze(), zt.squeeze()))
        ref = x["losses"][mode]
        assert abs(pl - ref[0]) < 2e-5 * abs(ref[0]) and abs(vl - ref[1]) < 2e-5 * max(abs(ref[1]), 1e-3), (mode, pl, vl, ref)
    return len(x["tuples"])
