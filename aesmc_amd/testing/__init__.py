"""Test support: contract-conforming models and RNG record/replay (not part of the hot path)."""
