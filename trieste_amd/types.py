"""Boundary types (reference trieste/types.py): arrays are numpy float64 on the host or torch
float64 CUDA tensors on the device; a Tag names a (model, dataset) pair."""
from typing import Any, Hashable

TensorType = Any
Tag = Hashable
