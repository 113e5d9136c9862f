"""Drop-in import name: ``import pysubstringsearch`` resolves to the MI355X
engine (pysubstringsearch_amd) with the reference's public names."""
from pysubstringsearch_amd import Reader, Writer  # noqa: F401

__all__ = ['Writer', 'Reader']
