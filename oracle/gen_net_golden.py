#!/usr/bin/env python3
"""oracle/gen_net_golden.py -- TEST INFRASTRUCTURE ONLY.

Second golden generator: everything that involves the reference's OWN network file
(/root/reference/src/py/net.py) and a few fixtures the first generator (gen_golden.py) did not
capture.  Runs only in the build container, imports the real reference (oracle/_ref/r{8,14} +
/root/reference/src/py) and writes DATA ONLY under tests/golden/:

  net_r{R}_b{B}_h{H}.npz   fp32 logits/value of the reference's ResNet(B blocks, H hidden), built under
                           torch.manual_seed(seed) with perturbed BatchNorm statistics, on 32 golden
                           positions encoded by the reference's GetEncodedState (1 024 sampled columns + row sum /
                           absmax / argmax of every row, and EVERY column of four rows); per-parameter checksums
                           so that a test can prove its own module holds the same weights
                           (the <= 1e-3 logits check against the reference's net.py)
  recnet_r{R}.npz          MCTS.search of the reference driven by the reference's ResNet with every
                           (input, logits, value) of the run RECORDED
  ref_extra_r{R}.json.gz   800-simulation searches (synthetic evaluators), all five start layouts as the
                           reference's fen_parser reads them (R = 14 run only), and one training batch
                           (encoded state, dense pi, z) with the reference-net loss of alphazero.py:181-209

    python oracle/gen_net_golden.py --size 8
    python oracle/gen_net_golden.py --size 14
"""
import argparse
import gzip
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402  (setup_imports: reference module path + line_profiler stub)

NET_CASES = {14: [(10, 128, 0), (20, 256, 0)], 8: [(4, 64, 0), (10, 128, 0), (15, 256, 0)]}   # (15, 256): the reference's shipped model, alphazero.py:288
N_POS = 32
N_IDX = 1024
FULL_ROWS = (0, 9, 18, 27)        # positions whose WHOLE logits row is stored (fp32; the others: N_IDX columns + a row sum)


def perturb_bn(model, seed):
    """non-trivial BatchNorm statistics so that BN folding is exercised (same rule as tests/test_nn_gpu.py)"""
    import torch
    g = torch.Generator().manual_seed(seed + 1)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)


def param_sums(model):
    names, sums = [], []
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            names.append(k)
            d = v.detach().double()
            sums.append([float(d.sum()), float(d.abs().sum())])
    return names, np.asarray(sums, np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, required=True, choices=[8, 14])
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma list of parts: net,recnet,extra")
    args = ap.parse_args()
    parts = set(args.only.split(",")) if args.only else {"net", "recnet", "extra"}
    gen_golden.setup_imports(args.size)

    import torch
    import torch.nn.functional as F
    import alphazero_cpp as az
    import start_fens
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    from mcts import MCTS
    import net as refnet                      # /root/reference/src/py/net.py

    torch.set_num_threads(8)
    R = az.Board.nRows()
    assert R == args.size
    RR = R * R
    A = az.Board.action_space_size
    with gzip.open(os.path.join(args.out, "ref_r%d.json.gz" % R), "rt") as f:
        gold = json.load(f)
    fen = gold["fen"]

    def board_from_snapshot(snap):
        l2p = {}
        for colour, col in enumerate(snap["pl"]):
            for sq, typ in col:
                l2p[az.BoardLocation(sq // R, sq % R)] = az.Piece(az.PlayerColor(colour), az.PieceType(typ))
        return FourPlayerChess(az.Player(az.PlayerColor(snap["turn"])), l2p)

    def lists(b):
        return [[[pp.GetLocation().GetRow() * R + pp.GetLocation().GetCol(), int(pp.GetPiece().GetPieceType())]
                 for pp in col] for col in b.GetPieces()]

    def snapshot(b):
        return {"turn": int(b.GetTurn().GetColor()), "pl": lists(b)}

    def enc_nonzero(t):
        return [int(i) for i in torch.nonzero(t.flatten()).flatten().tolist()]

    # ------------------------------------------------------------------ 1. logits of the reference's net.py
    if "net" in parts:
        pos = []
        for gi, game in enumerate(gold["playouts"]):
            for pi in range(0, len(game), 7):
                if "enc" in game[pi]:
                    pos.append((gi, pi))
        pos = pos[:N_POS]
        enc = torch.cat([az.Board.GetEncodedState(board_from_snapshot(gold["playouts"][gi][pi]["before"]), "cpu") for gi, pi in pos])
        for (gi, pi), e in zip(pos, enc):           # the positions ARE the golden ones
            assert enc_nonzero(e) == gold["playouts"][gi][pi]["enc"]
        idx = np.sort(np.random.default_rng(12345).choice(A, N_IDX, replace=False)).astype(np.int32)
        for blocks, hidden, seed in NET_CASES[R]:
            torch.manual_seed(seed)
            model = refnet.ResNet(FourPlayerChess, blocks, hidden, "cpu")
            perturb_bn(model, seed)
            model.eval()
            with torch.no_grad():
                lg, va = model(enc)
            names, sums = param_sums(model)
            path = os.path.join(args.out, "net_r%d_b%d_h%d.npz" % (R, blocks, hidden))
            full_rows = np.asarray([r for r in FULL_ROWS if r < len(pos)], np.int32)
            np.savez_compressed(path, pos=np.asarray(pos, np.int32), idx=idx, logits=lg[:, torch.from_numpy(idx.astype(np.int64))].numpy(),
                                full_rows=full_rows, full_logits=lg[torch.from_numpy(full_rows.astype(np.int64))].numpy(),
                                value=va.squeeze(1).numpy(), rowsum=lg.double().sum(dim=1).numpy(), absmax=lg.abs().max(dim=1).values.numpy(),
                                argmax=lg.argmax(dim=1).numpy().astype(np.int32), pnames=np.asarray(json.dumps(names)), psums=sums,
                                meta=np.asarray(json.dumps({"R": R, "blocks": blocks, "hidden": hidden, "seed": seed, "torch": torch.__version__,
                                                            "source": "reference src/py/net.py ResNet, eval(), fp32 CPU"})))
            print("wrote", path, os.path.getsize(path), "bytes; logit range", float(lg.abs().max()))
            del model, lg, va

    # ------------------------------------------------------------------ 2. recorded-network search
    if "recnet" in parts:
        torch.manual_seed(4)
        model = refnet.ResNet(FourPlayerChess, 2, 32, "cpu")
        perturb_bn(model, 4)
        model.eval()

        class Recorder:
            device = "cpu"

            def __init__(self):
                self.enc, self.logits, self.value, self.count = [], [], [], []

            def __call__(self, x):
                lg, v = model(x)
                self.count.append(int(x.shape[0]))
                for b in range(x.shape[0]):
                    self.enc.append(np.asarray(enc_nonzero(x[b]), np.int32))
                self.logits.append(lg.detach().numpy().copy())
                self.value.append(v.detach().numpy().reshape(-1).copy())
                return lg, v

        b0 = FourPlayerChess(*parse_board_args_from_fen(fen, R))
        states = [b0]
        sims = 32 if R == 8 else 10
        if R == 8:      # second game one ply further: different turn in the same batch (Q6)
            b1 = FourPlayerChess(*parse_board_args_from_fen(fen, R))
            flats = sorted(set(m.GetFlatIndex() for m in b1.GetLegalMoves()))
            b1 = b1.TakeAction(az.Move(flats[3]))
            states.append(FourPlayerChess(b1.GetTurn(), {pp.GetLocation(): pp.GetPiece() for col in b1.GetPieces() for pp in col}))
        before = [snapshot(s) for s in states]
        rec = Recorder()
        with torch.no_grad():
            roots = MCTS(FourPlayerChess, rec, {"pool_size": 10, "C": 3.0, "num_searches": sims}).search(states)
        out = []
        for r, s in zip(roots, states):
            out.append({"root_n": r.GetVisitCount(), "after": lists(s),
                        "children": [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount(),
                                      [[g.GetMoveMade().GetFlatIndex(), g.GetVisitCount()] for g in c.GetChildren()]]
                                     for c in r.GetChildren()]})
        enc_off = np.cumsum([0] + [len(e) for e in rec.enc]).astype(np.int64)
        path = os.path.join(args.out, "recnet_r%d.npz" % R)
        np.savez_compressed(path, count=np.asarray(rec.count, np.int32), enc_idx=np.concatenate(rec.enc), enc_off=enc_off,
                            logits=np.concatenate(rec.logits), value=np.concatenate(rec.value),
                            meta=np.asarray(json.dumps({"R": R, "sims": sims, "C": 3.0, "before": before, "roots": out,
                                                        "net": "reference net.py ResNet(2,32), seed 4, perturbed BN, eval()"})))
        print("wrote", path, os.path.getsize(path), "bytes;", len(rec.count), "evaluator calls")

    # ------------------------------------------------------------------ 3. extras
    if "extra" in parts:
        X = {"R": R}
        # 3a. 800-simulation searches through the reference's MCTS.search (synthetic evaluators of gen_golden.py)
        w11 = (torch.arange(24 * RR) % 11).to(torch.float32).view(1, 24, R, R)
        widx = ((torch.arange(24 * RR, dtype=torch.int64) * 2654435761) % (1 << 32)).view(1, 24, R, R)

        class Eval:
            device = "cpu"

            def __init__(self, kind):
                self.kind = kind

            def __call__(self, x):
                B = x.shape[0]
                if self.kind == "zero":
                    return torch.zeros(B, A), torch.zeros(B, 1)
                if self.kind == "ramp":
                    logits = (-(torch.arange(A) % 7).to(torch.float32) / 8).repeat(B, 1)
                    v = (((x * w11).sum(dim=(1, 2, 3)) % 5) - 2) / 4
                    return logits, v.view(B, 1)
                h = ((x.to(torch.int64) * widx).sum(dim=(1, 2, 3))) % (1 << 32)
                i = torch.arange(A, dtype=torch.int64).view(1, A)
                u = ((h.view(B, 1) * 2246822519 + i * 40503 + ((i * i) % 8191) * 69069) % (1 << 32)) >> 16
                return u.to(torch.float32) / 8192.0 - 4.0, (((h % 9).to(torch.float32) - 4) / 4).view(B, 1)

        searches = []
        for kind in (["hash", "ramp"] if R == 8 else ["zero", "hash"]):
            st = [FourPlayerChess(*parse_board_args_from_fen(fen, R))]
            before = [snapshot(s) for s in st]
            roots = MCTS(FourPlayerChess, Eval(kind), {"pool_size": 10, "C": 3.0, "num_searches": 800}).search(st)
            r = roots[0]
            searches.append({"kind": kind, "sims": 800, "C": 3.0, "before": before,
                             "roots": [{"root_n": r.GetVisitCount(), "after": lists(st[0]),
                                        "children": [[c.GetMoveMade().GetFlatIndex(), c.GetVisitCount(),
                                                      [[g.GetMoveMade().GetFlatIndex(), g.GetVisitCount()] for g in c.GetChildren()]]
                                                     for c in r.GetChildren()]}]})
            print("800-sim search", kind, "root_n", r.GetVisitCount())
        X["searches"] = searches

        # 3b. all five start layouts as fen_parser.parse_board_args_from_fen reads them
        if R == 14:
            layouts = {}
            for name, size in (("STANDARD", 14), ("THIRTEEN", 13), ("TEN", 10), ("EIGHT", 8), ("EIGHT_SIMPLE", 8)):
                f = getattr(start_fens, name).replace("\n", "")
                turn, l2p = parse_board_args_from_fen(f, size)[:2]
                layouts[name] = {"size": size, "turn": int(turn.GetColor()),
                                 "dict": [[k.GetRow() * size + k.GetCol(), int(v.GetColor()), int(v.GetPieceType())] for k, v in l2p.items()]}
            X["start_layouts"] = layouts

        # 3c. one training batch + the reference-net loss (alphazero.py:53-78 tuples, :181-209 loss)
        if R == 8:
            trace = gold["selfplay"][0]
            tuples = []
            for gidx, game in enumerate(trace["games"][:2]):
                st = FourPlayerChess(*parse_board_args_from_fen(fen, R))
                for ply, (mv, pi, z) in enumerate(zip(game["moves"], game["pi"], game["z"])):
                    if ply % 3 == 0 and len(tuples) < 16:
                        e = az.Board.GetEncodedState(st, "cpu").squeeze(0)     # per tuple, own rotation (alphazero.py:71-73)
                        tuples.append({"game": gidx, "ply": ply, "state": snapshot(st), "enc": enc_nonzero(e), "pi": pi, "z": z})
                    st = st.TakeAction(az.Move(mv))
            enc = torch.zeros(len(tuples), 24 * RR)
            pol = torch.zeros(len(tuples), A)
            for i, t in enumerate(tuples):
                enc[i, torch.tensor(t["enc"], dtype=torch.int64)] = 1.0
                for fl, n in t["pi"]:
                    pol[i, fl] = n
                pol[i] /= pol[i].sum()
            zt = torch.tensor([t["z"] for t in tuples], dtype=torch.float32).view(-1, 1)
            torch.manual_seed(9)
            model = refnet.ResNet(FourPlayerChess, 1, 64, "cpu")
            names, sums = param_sums(model)     # at initialisation
            losses = {}                         # train mode first (BN batch statistics; updates the running stats), then eval mode
            for mode in ("train", "eval"):
                model.train(mode == "train")
                with torch.no_grad():
                    out_policy, out_value = model(enc.view(-1, 24, R, R))
                    pl = F.cross_entropy(out_policy, pol)
                    vl = F.mse_loss(out_value.squeeze(), zt.squeeze())
                losses[mode] = [float(pl), float(vl), float(pl + vl)]
            X["train_batch"] = {"trace": 0, "tuples": tuples, "losses": losses, "seed": 9, "blocks": 1, "hidden": 64,
                                "pnames": names, "psums": sums.tolist()}
            print("train batch:", len(tuples), "tuples, losses", losses)
        path = os.path.join(args.out, "ref_extra_r%d.json.gz" % R)
        with gzip.open(path, "wt") as f:
            json.dump(X, f, separators=(",", ":"))
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
