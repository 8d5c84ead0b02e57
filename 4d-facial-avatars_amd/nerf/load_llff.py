"""Vanilla-NeRF LLFF loader of the reference (nerf/load_llff.py): out of the NeRFace hot-path scope.
The symbol exists because train_transformed_rays.py imports it (TR:17-21) without using it."""


def load_llff_data(*args, **kwargs):
    raise NotImplementedError("load_llff_data (vanilla NeRF datasets) is outside the NeRFace hot path and is not provided")
