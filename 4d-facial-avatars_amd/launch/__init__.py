"""Launchers around the MI355X `nerf` package: the reference's two caller loops as frame-sharded multi-GPU programs."""
