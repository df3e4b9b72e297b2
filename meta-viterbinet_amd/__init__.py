"""MI355X-native Viterbi / ViterbiNet detection engine: drop-in 'val' hot path for
tomerraviv95/meta-viterbinet (C ABI in include/mvn.h, kernels in csrc/mvn_hip.hip)."""
from . import _lib
from .channel import BPSKModulator, ReferenceWordStream, estimate_channel, generate_words, transmit
from .detectors import HIDDEN1_SIZE, HIDDEN2_SIZE, META_VNETDetector, VADetector, VNETDetector
from .ecc import rs_decode, rs_encode
from .harness import (data_indices, detect_by_word, eval_by_word, eval_counters, replica_eval, shard_range, sharded_eval,
                      single_eval_at_point, synthetic_words, va_monte_carlo)
from .meta import GraphedMetaStep, copy_model, meta_train_loop
from .online import OnlineTrainer
from .trials import TrialBank, TrialDraws, eval_by_word_batched
from .metrics import calculate_error_rates, count_errors, rates_from_counters
from .trellis import acs_block, acs_sweep, acs_sweep_survivors, calculate_states, create_transition_table, traceback

__all__ = [
    "VADetector", "VNETDetector", "META_VNETDetector", "HIDDEN1_SIZE", "HIDDEN2_SIZE",
    "create_transition_table", "acs_block", "acs_sweep", "acs_sweep_survivors", "traceback", "calculate_states",
    "calculate_error_rates", "count_errors", "rates_from_counters",
    "estimate_channel", "BPSKModulator", "transmit", "generate_words", "ReferenceWordStream", "rs_encode", "rs_decode", "OnlineTrainer", "meta_train_loop", "GraphedMetaStep", "copy_model",
    "shard_range", "data_indices", "synthetic_words", "va_monte_carlo", "eval_counters", "single_eval_at_point",
    "sharded_eval", "detect_by_word", "eval_by_word", "replica_eval", "TrialBank", "TrialDraws", "eval_by_word_batched",
]
