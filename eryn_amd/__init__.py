"""eryn_amd - MI355X-native stepping engine behind Eryn's EnsembleSampler / Move / State API
for the stretch-move + parallel-tempering path (hand-written HIP for gfx950 behind a C ABI)."""
__version__ = "0.1.0"
