"""Vanilla-NeRF Blender loader of the reference (nerf/load_blender.py): out of the NeRFace hot-path scope."""


def load_blender_data(*args, **kwargs):
    raise NotImplementedError("load_blender_data (vanilla NeRF datasets) is outside the NeRFace hot path and is not provided")
