"""MI355X-native mix-and-separate train step (package
``audio-visual-sepatation-in-visual-agnostic-situtation_amd``; ``import avsep_amd`` is an alias).

Layout: ``csrc/`` HIP kernels + C ABI (include/avsep.h) -> ``lib.py`` ctypes binding ->
``kernels.py`` tensor wrappers -> ``models/`` (ModelBuilder / activate surface of the reference's
models/__init__.py) and ``net_wrapper.py`` (NetWrapper / train_step surface of main.py).
"""
from . import lib, kernels, arguments, synth          # noqa: F401
from . import models, net_wrapper, dp, evaluate, bss_eval, sopp, inference, checkpoint, dataset, train  # noqa: F401
from .models import ModelBuilder, activate             # noqa: F401
from .net_wrapper import NetWrapper, create_optimizer, train_step, adjust_learning_rate  # noqa: F401
from .arguments import ArgParser                       # noqa: F401

__version__ = "0.1.0"
