"""golf_amd — MI355X-native GOLF time-varying LPC synthesis filter + glottal-flow source.

Drop-in nn.Modules (same class names / init_args / forward / .ctrl as the reference's
models/{filters,synth,noise,sf,hpn}.py) over hand-written HIP kernels for gfx950, bound through the
C ABI in include/golf_amd.h.  There is no CPU or PyTorch-eager fallback for the kernels.
"""
__version__ = "0.1.0"
