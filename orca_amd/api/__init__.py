"""Host-side mirror of the reference's Python entry points for the hot path: same
class / function names, argument meaning and error behaviour as LuckyLYM/Orca, implemented on the
C ABI of liborca_hip.so.  NumPy in / NumPy or torch out exactly where the reference does; all device
work is HIP (there is no CPU fallback).  Forward only: the reference's autograd path through these
modules is not implemented yet ("out of scope this round")."""
from .neighbor_finder import NeighborFinder, get_neighbor_finder            # utils/util.py:91-151
from .time_encoding import TimeEncode                                       # model/time_encoding.py
from .temporal_attention import TemporalAttentionLayer, MergeLayer          # model/temporal_attention.py
from .history import History                                                # modules/history.py
from .memory import Memory                                                  # modules/memory.py
from .memory_updater import get_memory_updater, GRUMemoryUpdater            # modules/memory_updater.py
from .message_aggregator import get_message_aggregator                      # modules/message_aggregator.py
from .embedding_module import get_embedding_module, GraphAttentionEmbedding  # modules/embedding_module.py
from .cache import get_cache_plan                                           # utils/cache.py
from .tgn_model import TGN                                                  # model/tgn_model.py (thin caller)
from .data import Data, get_data, load_feat                                 # utils/data_processing.py
from .evaluation import RandEdgeSampler, eval_edge_prediction               # utils/util.py:68-87, evaluation/evaluation.py
