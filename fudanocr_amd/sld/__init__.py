"""Stroke-level-decomposition transformer recognizer (BASELINE configs[4]) on the HIP kernels."""
