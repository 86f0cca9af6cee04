"""Package shim: with ``dropin/`` on ``sys.path`` ahead of the reference checkout, ``Utils.Loss``
(``Trainer.py:7``) resolves to the loss of this package, while every other ``Utils.*`` module
(``Utils.Camera``, ``Utils.DataLoader``, ``Utils.ContainerUtils`` ...) still comes from the reference: the
package's search path is this directory followed by every other ``Utils`` directory on ``sys.path``."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
__path__ = [_here]
for _entry in sys.path:
    _cand = os.path.join(os.path.abspath(_entry or os.getcwd()), "Utils")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != _here and _cand not in __path__:
        __path__.append(_cand)
