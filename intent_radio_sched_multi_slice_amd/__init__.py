"""MI355X-native batched RAN-slicing environment step."""
__version__ = "0.1.0"
