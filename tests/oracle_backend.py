"""Test helper: the CPU oracle dressed as the product's backends (oracle/backends.py)."""
from oracle.backends import OracleMuseSpectra, OracleSpectra, patch_neighbors  # noqa: F401
