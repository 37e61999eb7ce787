from .move import Move
from .stretch import StretchMove
from .tempering import TemperatureControl, make_ladder

__all__ = ["Move", "StretchMove", "TemperatureControl", "make_ladder"]
