from .registry import OpRegistry, OpResolver, SupportResult

__all__ = ["OpRegistry", "OpResolver", "SupportResult"]
