"""CPU oracle for the mix-and-separate train step.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference's
algorithm (abcqmars/audio-visual-sepatation-in-visual-agnostic-situtation) for
the hot path named in BASELINE.json.  Every function cites the reference
file:line it follows.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (the ``*_amd`` package) never imports it and has
no CPU fallback.

Parity pin: the reference holds no tests/golden vectors of its own
(SURVEY.md §4).  The oracle is pinned against outputs of the reference itself,
imported on CPU in the build container by ``oracle/gen_golden.py`` (recipe in
SURVEY.md §8(c)); the resulting vectors live in ``tests/golden/`` and
``tests/test_oracle_golden.py`` replays them without the reference present.
Third-party arithmetic outside the reference tree (torchvision resnet18,
librosa stft) is unpinned and restated from its published definition; see
DESIGN.md.
"""
from . import nets, criterion, step, stft  # noqa: F401
