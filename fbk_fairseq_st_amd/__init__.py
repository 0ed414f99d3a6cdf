"""MI355X-native (gfx950) speech-translation hot path behind the fairseq plug-in surface of
FBK-fairseq-ST's `conv_transformer`.  Importing the package does not touch the GPU.

The package is a fairseq *user directory*: `--user-dir <path>/fbk_fairseq_st_amd` makes fairseq import it
(fairseq/utils.py:344-359) and the imports below register the models / architectures, tasks and criteria of the S2T
path -- into fairseq's own registries when fairseq is the importing process (registry.py), into the package's
standalone ones otherwise -- as `examples/speech_recognition/__init__.py:1` does for the reference.
"""
__version__ = "0.2.0"

from . import tasks, criterions, conv_transformer  # noqa: E402,F401  (registration side effects)
from . import registry as _registry  # noqa: E402

if _registry.inside_fairseq():
    from . import fairseq_optim  # noqa: E402,F401  (`--optimizer adam` of fairseq's own trainer over the arena)
