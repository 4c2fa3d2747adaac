"""`MCTS` -- drop-in for the reference's src/py/mcts.py:8-89.

Same constructor and `search(games) -> root nodes` contract; the body (per-simulation Python loop
over ChooseLeaf / GetEncodedStates / softmax / mask / BackpropagateNodes / ExpandNodes) is replaced
by the engine:

* `neural_net` is a ResNet (net.py here, or any module with the reference's parameter names):
  its BN-folded weights are exported once per parameter version and the whole search runs on the
  GPU (`fpc_search_run`: select -> encode -> MFMA ResNet -> expand, no host round trips).
* `neural_net` is any other callable with a `.device` attribute (the reference's evaluator seam,
  mcts.py:65-66; synthetic / recorded evaluators in tests): the engine is driven step-wise and the
  callable sees the encoded leaves `[G,24,R,R]` and returns `(logits [G,A], value [G,1])`.
  Difference from the reference: the batch always has one slot per game (finished games are
  all-zero rows whose outputs are ignored) instead of only the live leaves.
"""
import numpy as np
import torch

import alphazero_cpp as az


def _param_version(model):
    return (id(model), sum(int(p._version) for p in model.parameters()), sum(int(b._version) for b in model.buffers()),
            bool(model.training))


class _DevPtr:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def _is_resnet(m):
    return isinstance(m, torch.nn.Module) and all(hasattr(m, k) for k in ("startBlock", "backBone", "policyHead", "valueHead"))


class MCTS:
    def __init__(self, gameType, neural_net, args):
        self.gameType = gameType
        self.args = args
        self.neural_net = neural_net
        self.nn_dtype = int(args.get("nn_dtype", 0)) if hasattr(args, "get") else 0
        # "full" (default): whole policy Linear + full softmax, the reference's arithmetic op for op;
        # "legal": opt-in legal-moves-only policy head (same priors up to f32 rounding, ~1.6x the
        # simulations/s at 14x14; include/fpc_engine.h fpc_set_policy_mode)
        self.policy_head = str(args.get("policy_head", "full")) if hasattr(args, "get") else "full"
        self._native = _is_resnet(neural_net)
        # N4 (explicitly NOT part of parity with the reference): args["rules"] = "strict"
        # (default: every quirk of the reference, bit-exact) or "fixed" (AlphaZero PUCT with the child's value
        # negated, per-sample rotation, un-shifted input planes, full moves in the tree), or an int of
        # fpc_ffi.RULES_* bits; args["root_noise"] = True applies Dirichlet(dirichlet_alpha) noise with weight
        # dirichlet_epsilon to the root priors inside the fused search loop, seeded by args["noise_seed"].
        rules = args.get("rules", "strict") if hasattr(args, "get") else "strict"
        self.rules = {"strict": 0, "fixed": 15}.get(rules, rules)
        self.root_noise = bool(args.get("root_noise", False)) if hasattr(args, "get") else False
        self._noise_rng = np.random.default_rng(int(args.get("noise_seed", 0))) if self.root_noise else None

    def sync_weights(self, eng):
        """(re-)export the network into the engine when its parameters changed (optimizer.step)."""
        import weights
        ver = _param_version(self.neural_net)
        if getattr(eng, "weights_version", None) != ver:
            eng.load_weights(weights.export_weights(self.neural_net, self.nn_dtype))   # the module stays where it is
            eng.weights_version = _param_version(self.neural_net)

    def add_dirichlet_noise(self, policy, device):
        """Counterpart of the reference's MCTS.add_dirichlet_noise (mcts.py:45-56): blends every row of
        a [B, A] policy with one Dirichlet(alpha) sample over the whole action space,
        (1 - eps) * policy + eps * noise, alpha/eps from args["dirichlet_alpha"/"dirichlet_epsilon"].
        As in the reference it is provided but never called by search()."""
        alpha, eps = float(self.args["dirichlet_alpha"]), float(self.args["dirichlet_epsilon"])
        n_actions = int(self.gameType.action_space_size)
        draw = np.random.dirichlet(np.full(n_actions, alpha), size=int(policy.shape[0]))
        noise = torch.as_tensor(draw, dtype=torch.float32, device=device)
        return policy * (1.0 - eps) + noise * eps

    @torch.no_grad()
    def search(self, games):
        G = len(games)
        sims = int(self.args["num_searches"])
        eng = az.engine(G, sims, self.nn_dtype if self._native else None)
        pods = [g._b for g in games]
        eng.set_rules(int(self.rules))
        if self.root_noise:
            import fpc_ffi
            gamma = self._noise_rng.standard_gamma(float(self.args["dirichlet_alpha"]), size=(G, fpc_ffi.MAX_MOVES)).astype(np.float32)
            eng.set_root_noise(gamma, float(self.args["dirichlet_epsilon"]))
        else:
            eng.set_root_noise(None, 0.0)
        eng.search_begin(pods, float(self.args["C"]))
        if self._native:
            self.sync_weights(eng)
            eng.set_policy_mode(self.policy_head == "legal")
            eng.search_run(sims)
        else:
            self._search_external(eng, G, sims)
        res = eng.search_results(roots=pods)          # root states keep the piece-list order the search left
        # the roots are views of the result arrays: child objects are made when GetChildren() is first called on a
        # root, TakeAction / GetGameResult of a root child come from one batched prefetch (alphazero_cpp._SearchBatch)
        batch = az._SearchBatch(eng, res, self.args["C"])
        roots = []
        for g, game in enumerate(games):
            root = az.Node(self.args["C"], game, visit_count=int(res["root_n"][g]))
            root._batch, root._g = batch, g
            game.SetRootNode(root)
            roots.append(root)
        assert int(res["n_children"].min()) > 0         # mcts.py:40-41
        return roots

    def _search_external(self, eng, G, sims):
        dev = str(getattr(self.neural_net, "device", "cpu"))
        on_gpu = dev.startswith("cuda") or dev == "gpu"
        host_engine = not torch.cuda.is_available()    # only true for the test-suite's emulator build
        R, A = eng.R, eng.A
        keep = None
        # mcts.py:36-38: select -> evaluate -> expand per simulation; between two evaluations the expansion and the
        # next selection are one launch (fpc_search_expand_select)
        n_live, enc_ptr = eng.search_select() if sims > 0 else (0, None)
        for i in range(sims):
            last = i == sims - 1
            if n_live == 0:
                if not last:
                    n_live, enc_ptr = eng.search_select()
                continue
            if host_engine:
                import ctypes
                import numpy as np
                enc = torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(enc_ptr, ctypes.POINTER(ctypes.c_float)),
                                                             shape=(G, 24, R, R)).copy())
                logits, value = self.neural_net(enc)
                logits = logits.to(torch.float32).contiguous().view(G, A)
                value = value.to(torch.float32).contiguous().view(G)
            else:
                enc = torch.as_tensor(_DevPtr(enc_ptr, (G, 24, R, R)), device="cuda")
                logits, value = self.neural_net(enc if on_gpu else enc.cpu())
                logits = logits.to(device="cuda", dtype=torch.float32).contiguous().view(G, A)
                value = value.to(device="cuda", dtype=torch.float32).contiguous().view(G)
                torch.cuda.synchronize()
            keep = (logits, value)
            if last:
                eng.search_expand(logits.data_ptr(), value.data_ptr())
            else:
                n_live, enc_ptr = eng.search_expand_select(logits.data_ptr(), value.data_ptr())
            if not host_engine:
                torch.cuda.synchronize()
        del keep
