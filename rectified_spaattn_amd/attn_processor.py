"""Collect / install attention processors on any module tree whose attention layers expose
get_processor / set_processor -- the reference's helper pair (attn_processor.py:6-62), which also works on
models that lack diffusers' mixin (Wan)."""
from typing import Dict, Union


def _walk(module, prefix=""):
    for name, child in module.named_children():
        path = f"{prefix}.{name}" if prefix else name
        yield path, child
        yield from _walk(child, path)


def get_attn_processors(module) -> Dict[str, object]:
    """{"<path>.processor": processor} for every sub-module with a get_processor() method."""
    return {f"{path}.processor": m.get_processor() for path, m in _walk(module) if hasattr(m, "get_processor")}


def set_attn_processor(module, processor: Union[object, Dict[str, object]]):
    """Install one processor everywhere, or a dict keyed like get_attn_processors(); a dict of the wrong size
    raises ValueError (attn_processor.py:45-49).  Dict entries are consumed (popped) as in the reference."""
    count = len(get_attn_processors(module))
    if isinstance(processor, dict) and len(processor) != count:
        raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not "
                         f"match the number of attention layers: {count}. Please make sure to pass {count} "
                         f"processor classes.")
    for path, m in _walk(module):
        if hasattr(m, "set_processor"):
            m.set_processor(processor.pop(f"{path}.processor") if isinstance(processor, dict) else processor)
