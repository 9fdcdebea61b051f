"""Network adapter contract — the reference's ``cobel.network.network.Network`` ABC
(network/network.py:12-147).  Agents exchange NumPy batches with it; the PyTorch adapter adds
device-tensor entry points so the vectorised DQN never leaves the GPU."""
from __future__ import annotations

import abc


class Network(abc.ABC):
    @abc.abstractmethod
    def predict_on_batch(self, batch):
        ...

    @abc.abstractmethod
    def train_on_batch(self, batch, targets):
        ...

    @abc.abstractmethod
    def get_weights(self):
        ...

    @abc.abstractmethod
    def set_weights(self, weights) -> None:
        ...

    @abc.abstractmethod
    def clone(self):
        ...

    @abc.abstractmethod
    def set_optimizer(self, optimizer, parameters=None) -> None:
        ...

    @abc.abstractmethod
    def set_loss(self, loss, parameters=None) -> None:
        ...

    @abc.abstractmethod
    def get_layer_activity(self, batch, layer):
        ...

    @abc.abstractmethod
    def set_trainable(self, layers, trainable) -> None:
        ...
