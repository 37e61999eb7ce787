from .move import Move
from .device import DeviceMove
from .stretch import StretchMove
from .gaussian import GaussianMove, MHMove
from .tempering import TemperatureControl, make_ladder

__all__ = ["Move", "DeviceMove", "StretchMove", "GaussianMove", "MHMove", "TemperatureControl", "make_ladder"]
