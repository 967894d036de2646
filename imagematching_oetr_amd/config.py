"""Configuration node for the OETR overlap estimator.

The reference builds its model from a yacs ``CfgNode`` (reference
``src/config/default.py:3-76``).  yacs is not a dependency here: ``Cfg`` is a
small attribute-dict that offers the handful of operations callers of the
reference actually use (attribute access, item access, ``clone``), and
``get_cfg_defaults`` returns the same tree of constants the model reads:
``OETR.MODEL``, ``OETR.NORM_INPUT``, ``OETR.BACKBONE.{NUM_LAYERS,LAYER,
LAST_LAYER,STRIDE}``, ``OETR.NECK.MAX_SHAPE`` and ``OETR.LOSS.{OIOU,
CYCLE_OVERLAP}``.  Dataset/training keys are out of scope.
"""
import copy


class Cfg(dict):
    """dict with attribute access; nested dicts are converted on assignment."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, Cfg):
            value = Cfg(value)
        super().__setitem__(key, value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def clone(self):
        return copy.deepcopy(self)


def get_cfg_defaults():
    """Fresh copy of the defaults (reference ``src/config/default.py:72-76``)."""
    return Cfg(
        OUTPUT='',
        OETR=dict(
            CHECKPOINT=None,
            BACKBONE_TYPE='ResNet',
            MODEL='oetr',
            NORM_INPUT=True,
            BACKBONE=dict(NUM_LAYERS=50, STRIDE=16, LAYER='layer3',
                          LAST_LAYER=1024),
            NECK=dict(D_MODEL=256, LAYER_NAMES=['self', 'cross'] * 4,
                      ATTENTION='linear', MAX_SHAPE=(100, 100)),
            HEAD=dict(D_MODEL=256, NORM_REG_TARGETS=True),
            LOSS=dict(OIOU=False, CYCLE_OVERLAP=False, FOCAL_ALPHA=0.25,
                      FOCAL_GAMMA=2.0, REG_WEIGHT=1.0, CENTERNESS_WEIGHT=1.0),
        ),
    )
