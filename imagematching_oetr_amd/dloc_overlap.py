"""Plug-in wrapper with the interface of the reference's
``dloc/core/overlaps/oetr.py:15-46`` (``BaseModel`` contract,
``dloc/core/utils/base_model.py:8-34``): ``OETR(conf, model_path)(data)`` ->
``(box1, box2)`` with ``data = {'image0': ..., 'image1': ...}``.

Two ways in:

* stand-alone: :class:`OETR` below (its base class restates the ``BaseModel``
  contract, so nothing of the reference needs to be importable);
* inside the reference tree: ``dynamic_load`` (``base_model.py:37-46``) only accepts
  classes DEFINED in ``dloc/core/overlaps/<name>.py`` that subclass the reference's own
  ``BaseModel``; the three-line shim INTEGRATION.md shows does that by combining
  :class:`OETRPluginMixin` with the reference's base class
  (``tests/test_dloc_plugin_cpu.py`` runs that shim through a restated ``dynamic_load``).
"""
from copy import copy
from pathlib import Path

import torch
from torch import nn

from .config import get_cfg_defaults
from .model import build_detectors


class OETRPluginMixin:
    """``_init`` / ``_forward`` of the overlap estimator (reference ``oetr.py:28-46``)."""

    default_conf = {
        'model': 'oetr',
        'num_layers': 50,
        'stride': 32,
        'last_layer': 1024,
        'weights': 'oetr.pth',
    }
    required_data_keys = ['image0', 'image1']

    def build_cfg(self, conf):
        cfg = get_cfg_defaults()
        cfg.OETR.MODEL = conf['model']
        cfg.OETR.BACKBONE.STRIDE = conf['stride']
        cfg.OETR.BACKBONE.LAYER = conf['layer']        # KeyError if absent, as in the reference
        cfg.OETR.BACKBONE.LAST_LAYER = conf['last_layer']
        return cfg

    def _init(self, conf, model_path):
        self.conf = {**self.default_conf, **conf}
        self.cfg = self.build_cfg(self.conf)
        self.net = build_detectors(self.cfg.OETR)
        if model_path is not None:                     # (None: random init, for tests / benches)
            model_file = Path(model_path) / self.conf['weights']
            self.net.load_state_dict(torch.load(model_file, map_location='cpu'))   # strict

    def _forward(self, data):
        box1, box2 = self.net.forward_dummy(data['image0'], data['image1'])
        # the dloc pipeline reads the boxes right away (Matching.forward, evaluation.py:77-113):
        # settle the range check of THIS pair before handing them over (the batched front-end,
        # pipeline.forward_pairs, defers it to the next batch instead)
        self.net.hip_flush()
        return box1, box2


class _BaseModel(nn.Module):
    """Restatement of the contract of reference ``base_model.py:8-34``."""

    default_conf = {}
    required_data_keys = []

    def __init__(self, conf, model_path=None):
        super().__init__()
        self.conf = conf = {**self.default_conf, **conf}
        self.required_data_keys = copy(self.required_data_keys)
        self._init(conf, model_path)
        self.model_path = model_path

    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, 'Missing key {} in data'.format(key)
        return self._forward(data)


class OETR(OETRPluginMixin, _BaseModel):
    pass
