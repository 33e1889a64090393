from .capture import CapturedRender
from .graph import render_grafx
from .order.graph import compute_render_order, reorder_for_fast_render
from .prepare import RenderData, prepare_render
