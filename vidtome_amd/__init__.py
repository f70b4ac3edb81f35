"""vidtome_amd -- MI355X-native implementation of VidToMe's cross-frame token-merging hot path.

Same public surface as the reference's ``vidtome`` package (vidtome/__init__.py:1-4):
``apply_patch, remove_patch, update_patch, collect_from_patch`` plus the ``merge`` and ``patch`` modules.
Everything executes in libvidtome_hip.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from . import merge, patch
from .patch import apply_patch, remove_patch, update_patch, collect_from_patch

__all__ = ["merge", "patch", "apply_patch", "remove_patch", "update_patch", "collect_from_patch"]
