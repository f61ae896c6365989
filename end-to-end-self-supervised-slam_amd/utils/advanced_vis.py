"""Import-surface stand-in for utils/advanced_vis.py (online_adaption.py:25 imports plotly_map_update_visualization;
only demo.py:254 calls it -- OUT OF SCOPE).  Maps are exported with utils.export.save_ply."""

_MSG = "out of scope: SURVEY.md section 2 row P10 -- plotly map animation; use utils.export.save_ply for the fused map"


def plotly_map_update_visualization(*args, **kwargs):
    raise NotImplementedError(_MSG)
