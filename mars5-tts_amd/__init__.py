"""MI355X-native engine for the MARS5-TTS hot path (AR decode loop + multinomial-DDPM
NAR refinement) behind the reference's ``Mars5TTS.tts()`` / ``InferenceConfig`` surface.

Layout: ``csrc/`` hand-written HIP kernels for gfx950 + the C-ABI (``include/mars5_hip.h``),
host-side mirrors of the reference seams (``ar_generate``, ``diffuser``, ``model``,
``minbpe``), utterance sharding (``sharding``).  See DESIGN.md.
"""
__version__ = "0.1.0"
