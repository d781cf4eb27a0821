"""Interface stub (see tests/stubs/README.md): gym 0.18's spaces package layout (gym/spaces/box.py holds Box)."""
from .box import Box, Space          # noqa: F401
