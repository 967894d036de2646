"""Plug-in wrapper with the interface of the reference's
``dloc/core/overlaps/oetr.py:15-46`` (``BaseModel`` contract,
``dloc/core/utils/base_model.py:8-34``): ``OETR(conf, model_path)(data)`` ->
``(box1, box2)`` with ``data = {'image0': ..., 'image1': ...}``.

INTEGRATION.md shows the two-line change that makes the reference's
``dynamic_load(overlaps, 'oetr')`` pick this class up.
"""
from copy import copy
from pathlib import Path

import torch
from torch import nn

from .config import get_cfg_defaults
from .model import build_detectors


class OETR(nn.Module):
    default_conf = {
        'model': 'oetr',
        'num_layers': 50,
        'stride': 32,
        'last_layer': 1024,
        'weights': 'oetr.pth',
    }
    required_data_keys = ['image0', 'image1']

    def __init__(self, conf, model_path=None):
        super().__init__()
        self.conf = conf = {**self.default_conf, **conf}
        self.required_data_keys = copy(self.required_data_keys)
        self.model_path = model_path
        cfg = get_cfg_defaults()
        cfg.OETR.MODEL = conf['model']
        cfg.OETR.BACKBONE.NUM_LAYERS = conf['num_layers']
        cfg.OETR.BACKBONE.STRIDE = conf['stride']
        cfg.OETR.BACKBONE.LAYER = conf['layer']        # KeyError if absent, as in the reference
        cfg.OETR.BACKBONE.LAST_LAYER = conf['last_layer']
        self.cfg = cfg
        self.net = build_detectors(cfg.OETR)
        if model_path is not None:
            state = torch.load(Path(model_path) / conf['weights'],
                               map_location='cpu')
            self.net.load_state_dict(state)            # strict, as in the reference

    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, 'Missing key {} in data'.format(key)
        return self.net.forward_dummy(data['image0'], data['image1'])
