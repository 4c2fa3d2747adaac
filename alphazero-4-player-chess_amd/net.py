"""Training-side PyTorch module of the policy/value network -- the mirror of the reference's
src/py/net.py:6-63 (same architecture, same parameter names, so state_dicts are interchangeable).
PyTorch-ROCm is used for the optimiser step only; self-play inference runs in the engine's MFMA
kernels from an exported weight blob (weights.py)."""
import torch.nn.functional as F
from torch import nn


class ResBlock(nn.Module):            # net.py:49-63
    def __init__(self, num_hidden):
        super().__init__()
        self.conv1 = nn.Conv2d(num_hidden, num_hidden, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(num_hidden)
        self.conv2 = nn.Conv2d(num_hidden, num_hidden, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(num_hidden)

    def forward(self, x):
        residual = x
        x = F.relu(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        x += residual
        return F.relu(x)


class ResNet(nn.Module):              # net.py:6-46
    def __init__(self, gameType, num_resBlocks, num_hidden, device):
        super().__init__()
        self.device = device
        self.num_resBlocks, self.num_hidden = num_resBlocks, num_hidden
        self.board_size = gameType.nRows()
        n_state, n_act = gameType.num_state_channels, gameType.num_action_channels
        self.startBlock = nn.Sequential(nn.Conv2d(n_state, num_hidden, kernel_size=3, padding=1),
                                        nn.BatchNorm2d(num_hidden), nn.ReLU())
        self.backBone = nn.ModuleList([ResBlock(num_hidden) for _ in range(num_resBlocks)])
        self.policyHead = nn.Sequential(nn.Conv2d(num_hidden, n_act, kernel_size=3, padding=1),
                                        nn.BatchNorm2d(n_act), nn.ReLU(), nn.Flatten(),
                                        nn.Linear(gameType.action_space_size,
                                                  n_act * gameType.nRows() * gameType.nCols()))
        self.valueHead = nn.Sequential(nn.Conv2d(num_hidden, n_state, kernel_size=3, padding=1),
                                       nn.BatchNorm2d(n_state), nn.ReLU(), nn.Flatten(),
                                       nn.Linear(gameType.state_space_size, 1), nn.Tanh())
        self.to(device)

    def forward(self, x):
        x = self.startBlock(x)
        for resBlock in self.backBone:
            x = resBlock(x)
        return self.policyHead(x), self.valueHead(x)
