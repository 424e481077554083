from .KD import DistillKL

__all__ = ["DistillKL"]
