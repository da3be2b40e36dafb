"""MI355X-native data-parallel training framework.

Capabilities of the reference script (``/root/reference/main.py`` +
``README.md``), re-designed for gfx950: one process per GPU, RCCL over xGMI,
hand-written HIP kernels for the hot ops, a C++ gradient reducer.

Sub-packages
------------
``launch``    ``python -m pytorch_distributed_training_amd.launch`` (README.md:14,28,34)
``dist``      env:// process-group bootstrap, ``reduce_loss`` (main.py:16-20,34-37)
``data``      DistributedSampler, synthetic / CIFAR datasets, pinned prefetch loader (main.py:42-63)
``models``    ResNet-18/50/152, ViT-B/16 with torchvision ``state_dict`` keys (main.py:40)
``nn``        Conv/BN/SyncBN/loss modules that run on the HIP kernels (main.py:79-82)
``ops``       Python bindings to the HIP kernels + the pure-PyTorch oracles
``parallel``  DistributedDataParallel + reducer (main.py:83)
``optim``     fused Adam/AdamW/SGD over flat buffers (main.py:80)
``utils``     checkpoint, profiler/roctx, TSV/JSON metrics (main.py:65-78,105-117)
"""

__version__ = "0.1.0"

from . import dist  # noqa: F401
from .utils.env import native_available, require_native  # noqa: F401

__all__ = ["dist", "native_available", "require_native", "__version__"]
