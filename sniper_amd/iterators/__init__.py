"""Training / inference iterators with the reference's lib/iterators contract, GPU data path."""
from .MNIteratorE2E import MNIteratorE2E  # noqa: F401
