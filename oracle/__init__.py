"""CPU oracle for the ViT + DINO multi-crop training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.

Parity status: **unpinned at the timm / DINO-repo boundary** -- the reference
ships no tests, fixtures or golden vectors, its model code exists only as
CPython-3.7 bytecode that nothing imports, and timm / torchvision are absent
from this image.  The restatement is therefore pinned
against independent implementations that ARE installed (torch.nn.LayerNorm /
MultiheadAttention-free SDPA / Conv2d / weight_norm and a locally configured
HF ``ViTModel``), see ``tests/test_oracle.py``, and against the committed
fixtures in ``tests/golden/`` which were produced by ``oracle/make_golden.py``.
"""
