"""cmr-landmark-detection_amd: the MI355X-native training path of the RVIP heat-map U-Net.

Drop-in for ONE path of Cardio-AI/cmr-landmark-detection -- ``get_model(config)`` / ``create_unet(config)``
(src/models/Unets.py) and the ``model.fit`` step the Train notebooks / train_model.py run -- behind the same
config-dict, generator and callback protocols.  Compute = hand-written HIP kernels for gfx950 in
``csrc/`` behind the C ABI of ``include/rvip_hip.h``; this package is the host side.

The directory name carries a hyphen (it is the reference's name + ``_amd``); import it as
``import cmr_landmark_detection_amd as rvip`` (alias module at the repo root) or with importlib.
"""
from .Unets import create_unet, get_model, UnetPlan            # noqa: F401
from .keras_model import Model                                  # noqa: F401
from .ModelUtils import get_optimizer, Adam                     # noqa: F401
from . import Loss_and_metrics, Generators, KerasCallbacks, Preprocess      # noqa: F401
from . import _native                                           # noqa: F401

__all__ = ['create_unet', 'get_model', 'UnetPlan', 'Model', 'get_optimizer', 'Adam', 'Loss_and_metrics', 'Generators',
           'KerasCallbacks', 'Preprocess']
