"""Type stubs of the drop-in import name.  The reference ships the same surface as
pysubstringsearch/pysubstringsearch.pyi:1-44 beside py.typed; the classes here are the
MI355X engine's (pysubstringsearch_amd), whose extra keyword-only arguments and
extension methods are typed in pysubstringsearch_amd/__init__.pyi."""
from pysubstringsearch_amd import Reader as Reader
from pysubstringsearch_amd import Writer as Writer

__all__ = ['Writer', 'Reader']
