"""diff-sampler_amd: MI355X-native diffusion ODE-sampling engine (hot path of zju-pi/diff-sampler).

Import as ``diff_sampler_amd`` (see the shim ``diff_sampler_amd.py`` at the repo root).
Sub-modules keep the reference's module names so that they drop in:
``solvers``, ``solver_utils``, ``solvers_amed``, ``sample`` -- plus ``arch``/``engine`` (denoiser plan) and
``_lib`` (ctypes binding of the C-ABI library ``csrc/libdsamd.so``).
"""
__version__ = "0.6.0"
