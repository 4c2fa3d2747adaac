"""Training driver -- counterpart of the reference's AlphaZero class (src/py/alphazero.py:19-277)
without its wandb / pygame side channels.  Self-play runs on the engine (selfplay.play over
MCTS.search, whole search on the GPU when the model is a ResNet); the optimiser step stays plain
PyTorch(-ROCm):  loss = cross_entropy(policy_logits, pi) + mse(value, z)   (alphazero.py:200-209).

    az = AlphaZero(model, optimizer, FourPlayerChess, args, game_init_args)
    az.learn()

`args` takes the reference's keys (alphazero.py:291-306): max_game_length, C, num_searches,
num_iterations, num_games, num_parallel_games, batch_size, temperature, heuristic_weight,
replay_buffer_capacity, validation_buffer_capacity.
"""
import numpy as np
import torch
import torch.nn.functional as F

import alphazero_cpp as az
import selfplay
import tuples
from mcts import MCTS
from replay_buffer import ReplayBuffer


class AlphaZero:
    def __init__(self, model, optimizer, gameType, args, game_init_args=None, evaluator=None, seed=None):
        self.model, self.optimizer, self.gameType, self.args = model, optimizer, gameType, args
        self.game_init_args = game_init_args
        self.scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=1000, gamma=0.1)   # alphazero.py:25-27
        self.mcts = MCTS(gameType, evaluator if evaluator is not None else model, args)
        self.experience_buffer = ReplayBuffer(args["replay_buffer_capacity"])
        self.validation_buffer = ReplayBuffer(args["validation_buffer_capacity"])
        self.gen = torch.Generator().manual_seed(seed) if seed is not None else None
        self.history = []

    # ---- self-play (alphazero.py:81-178) ----
    def _new_game(self):
        return self.gameType() if not self.game_init_args else self.gameType(*self.game_init_args)

    def play(self):
        G = int(self.args["num_parallel_games"])
        games = [self._new_game() for _ in range(G)]
        # same (games, sims, dtype) request as MCTS.search makes, so the handle is not re-created under us
        eng = az.engine(G, int(self.args["num_searches"]), self.mcts.nn_dtype if self.mcts._native else None)
        L = int(self.args["max_game_length"])
        uniforms = torch.rand(L, G, generator=self.gen, dtype=torch.float64).tolist()

        def search_fn(pods):
            boards = [self.gameType._wrap(p) for p in pods]
            roots = self.mcts.search(boards)
            arrs = [r.child_arrays() for r in roots]                 # no Python object per child
            n = max(len(f) for f, _ in arrs)
            res = {"n_children": np.array([len(f) for f, _ in arrs]),
                   "flat": np.zeros((len(roots), n), np.int64), "visits": np.zeros((len(roots), n), np.int64)}
            for i, (f, v) in enumerate(arrs):
                res["flat"][i, :len(f)] = f
                res["visits"][i, :len(v)] = v
            return res

        episodes = selfplay.play(search_fn, eng, [g._b for g in games], self.args, uniforms)
        split = self.args["replay_buffer_capacity"] / (self.args["replay_buffer_capacity"] + self.args["validation_buffer_capacity"])
        for ep in episodes:                               # handle_terminal_state, alphazero.py:53-78
            for (pod, flats, visits), z in zip(ep.entries, ep.z):
                buf = self.experience_buffer if torch.rand(1, generator=self.gen).item() < split else self.validation_buffer
                buf.add((pod, flats, visits, float(z)))
        return episodes

    # ---- optimiser step (alphazero.py:181-258) ----
    def _batch(self, sample):
        eng = az.engine()
        A = self.gameType.action_space_size
        dev = next(self.model.parameters()).device
        # GetEncodedState(entry.state) encodes each tuple on its own => per-sample rotation (alphazero.py:71-73)
        x = np.concatenate([eng.encode([pod]) for pod, _, _, _ in sample])
        pi = torch.stack([tuples.dense_pi({"flat": np.asarray(f, np.int64), "visits": np.asarray(v, np.int64)}, A) for _, f, v, _ in sample])
        z = torch.tensor([s[3] for s in sample], dtype=torch.float32).view(-1, 1)
        return torch.from_numpy(x).to(dev), pi.to(dev), z.to(dev)

    def _loss(self, sample):
        x, pi, z = self._batch(sample)
        out_policy, out_value = self.model(x)
        policy_loss = F.cross_entropy(out_policy, pi)
        value_loss = F.mse_loss(out_value.squeeze(), z.squeeze())
        return policy_loss, value_loss

    def train(self):
        bs = int(self.args["batch_size"])
        if len(self.experience_buffer) < bs:
            return None
        last = None
        for _ in range(0, len(self.experience_buffer), bs):
            policy_loss, value_loss = self._loss(self.experience_buffer.sample(bs))
            loss = policy_loss + value_loss
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            self.scheduler.step()
            last = {"policy_loss": policy_loss.item(), "value_loss": value_loss.item(), "loss": loss.item()}
            self.history.append(last)
        return last

    @torch.no_grad()
    def validate(self):
        bs = int(self.args["batch_size"])
        if len(self.validation_buffer) < bs:
            return None
        policy_loss, value_loss = self._loss(self.validation_buffer.sample(bs))
        return {"policy_loss": policy_loss.item(), "value_loss": value_loss.item(), "loss": (policy_loss + value_loss).item()}

    def learn(self):                                      # alphazero.py:260-277
        for _ in range(int(self.args["num_iterations"])):
            self.model.eval()
            for _ in range(int(self.args["num_games"]) // int(self.args["num_parallel_games"])):
                self.play()
            self.model.train()
            self.train()
            self.model.eval()
            self.play()
            self.validate()
