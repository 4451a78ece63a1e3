"""picopose_amd — MI355X-native (gfx950) implementation of the PicoPose
three-stage correspondence hot path behind the reference's Python API."""

__version__ = "0.1.0"
