"""Model-surgery helpers around the quantised layers (reference: QuantTorch/utils/)."""
from .convertor import (convert, binary_net_convert, ternary_net_convert, dorefa_net_convert,
                        xnor_net_convert, log_lin_net_convert)
from .tools import flat_net
from .packed_state import packed_state_dict, load_packed_state_dict, packed_state_nbytes
from .graphs import AutoGraphed, GraphedModule, GraphedTrainStep, auto_graphed, graphed
from .implicit import implicit_graphs, implicit_graphs_off, implicit_graph_stats
from .data_parallel import GradientSynchronizer, clamp_weights_, broadcast_parameters

__all__ = ["convert", "binary_net_convert", "ternary_net_convert", "dorefa_net_convert",
           "xnor_net_convert", "log_lin_net_convert", "flat_net", "packed_state_dict", "load_packed_state_dict",
           "packed_state_nbytes", "GraphedModule", "GraphedTrainStep", "graphed", "AutoGraphed", "auto_graphed", "implicit_graphs", "implicit_graphs_off",
           "implicit_graph_stats", "GradientSynchronizer", "clamp_weights_",
           "broadcast_parameters"]
