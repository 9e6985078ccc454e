"""MI355X (gfx950) implementation of the Cognitive-Aim depth forward pass.

Public surface mirrors the reference's ``src/model.py``:
``create_model``, ``CognitiveAimModel``.
"""
from .model import CognitiveAimModel, create_model  # noqa: F401

__all__ = ["CognitiveAimModel", "create_model"]
