"""Drop-in `window_ann` module (reference: python_bindings/python_bindings.cpp, module name at
:160; imported star-wise by experiments/wrapper.py:1).  Backed by the MI355X engine."""
from rangefilteredann_amd._window_ann import *  # noqa: F401,F403
from rangefilteredann_amd._window_ann import defaults  # noqa: F401
