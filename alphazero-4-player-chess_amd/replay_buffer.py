"""Ring replay buffer of training tuples (counterpart of the reference's src/py/replay_buffer.py:4-20:
fixed capacity, overwrite-oldest, uniform sampling without replacement)."""
import random


class ReplayBuffer:
    def __init__(self, capacity, rng=None):
        self.capacity = int(capacity)
        self._items = []
        self._next = 0
        self._rng = rng or random

    def add(self, experience):
        if len(self._items) < self.capacity:
            self._items.append(experience)
        else:
            self._items[self._next] = experience
        self._next = (self._next + 1) % self.capacity

    def sample(self, batch_size):
        return self._rng.sample(self._items, batch_size)

    def __len__(self):
        return len(self._items)
