"""emdenoise: MI355X-native (gfx950) implementation of the CNN micrograph-denoising hot path of
Jeffrey-Ede/AI-CV-Automation-Elect-Micr, behind the reference's own Python call surface.

Python here is host logic only (shapes, buffers, streams); all arithmetic runs in hand-written
HIP kernels reached through the C ABI of libemdenoise.so (include/emdenoise.h).
"""
from . import _lib  # noqa: F401
from .kernel_denoiser import KernelParams, Micrograph_Autoencoder, kernel_denoise  # noqa: F401
from . import autoencoder, denoiser, gan, graphed, input_pipeline, kernel_denoiser, ops, streams, tf_checkpoint, train_ops, trainer, xception  # noqa: F401
from .denoiser import Denoiser, DenoiserEngine, architecture, synthetic_weights  # noqa: F401
from .trainer import DenoiserTrainer, get_model_fn  # noqa: F401

__all__ = ["KernelParams", "Micrograph_Autoencoder", "kernel_denoise"]
