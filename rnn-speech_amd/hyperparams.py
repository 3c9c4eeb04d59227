"""config.ini reader + checkpoint-side pickle logic (reference:
/root/reference/util/hyperparams.py:17-141).  Same section/key names and defaults; a few
optional keys are added for the MI355X build (n_mfcc, sample_rate)."""
import configparser
import logging
import os
import pickle
import time

_ACOUSTIC, _GENERAL, _TRAINING, _LOGGING = "acoustic_network_params", "general", "training", "logging"
# a change of any of these makes an existing checkpoint unusable (the reference compares the first four,
# util/hyperparams.py:75-92; n_mfcc / sample_rate are this build's extra keys and change the input layer's
# shape / the features' meaning)
_STRUCTURAL = ("num_layers", "hidden_size", "signal_processing", "language", "n_mfcc", "sample_rate", "bidirectional")
_STRUCTURAL_DEFAULTS = {"signal_processing": "mfcc", "language": "", "n_mfcc": 20, "sample_rate": 22050, "bidirectional": False}


def read_config_file(config_file):
    cp = configparser.ConfigParser()
    cp.read(config_file)
    d = {}
    for key, getter in (("num_layers", cp.getint), ("hidden_size", cp.getint),
                        ("dropout_input_keep_prob", cp.getfloat), ("dropout_output_keep_prob", cp.getfloat),
                        ("batch_size", cp.getint), ("mini_batch_size", cp.getint),
                        ("learning_rate", cp.getfloat), ("lr_decay_factor", cp.getfloat),
                        ("grad_clip", cp.getint), ("signal_processing", cp.get), ("language", cp.get),
                        ("rnn_state_reset_ratio", cp.getfloat)):
        d[key] = getter(_ACOUSTIC, key)
    d["use_config_file_if_checkpoint_exists"] = cp.getboolean(_GENERAL, "use_config_file_if_checkpoint_exists")
    d["steps_per_checkpoint"] = cp.getint(_GENERAL, "steps_per_checkpoint")
    d["steps_per_evaluation"] = cp.getint(_GENERAL, "steps_per_evaluation")
    d["checkpoint_dir"] = cp.get(_GENERAL, "checkpoint_dir")
    d["training_dataset_dirs"] = cp.get(_TRAINING, "training_dataset_dirs")
    d["training_filelist_cache"] = cp.get(_TRAINING, "training_filelist_cache", fallback=None)
    d["test_dataset_dirs"] = cp.get(_TRAINING, "test_dataset_dirs", fallback=None)
    d["train_frac"] = cp.getfloat(_TRAINING, "train_frac", fallback=None)
    d["max_input_seq_length"] = cp.getint(_TRAINING, "max_input_seq_length")
    d["max_target_seq_length"] = cp.getint(_TRAINING, "max_target_seq_length")
    tb = cp.get(_TRAINING, "tensorboard_dir", fallback=None)
    d["tensorboard_dir"] = tb if tb is not None and os.path.exists(tb) else None
    d["batch_normalization"] = cp.getboolean(_TRAINING, "batch_normalization", fallback=False)
    d["dataset_size_ordering"] = cp.get(_TRAINING, "dataset_size_ordering", fallback="False")
    d["log_file"] = cp.get(_LOGGING, "log_file", fallback=None)
    level = cp.get(_LOGGING, "log_level", fallback="WARNING")
    d["log_level"] = getattr(logging, level, None)
    if not isinstance(d["log_level"], int):
        raise ValueError("Invalid log level: %s" % level)
    # MI355X-build extras (absent from the reference's config.ini -> reference behaviour)
    d["n_mfcc"] = cp.getint(_ACOUSTIC, "n_mfcc", fallback=20)
    d["prefetch_batches"] = cp.getint(_TRAINING, "prefetch_batches", fallback=2)    # decode-ahead depth
    d["feature_cache_mb"] = cp.getint(_TRAINING, "feature_cache_mb", fallback=0)    # host feature cache, 0 = off
    d["precision"] = cp.get(_ACOUSTIC, "precision", fallback="f32")       # f32 (exact) | bf16x3 (split MFMA) | bf16 (plain bf16 operands)
    d["sample_rate"] = cp.getint(_TRAINING, "sample_rate", fallback=22050)
    d["bidirectional"] = cp.getboolean(_ACOUSTIC, "bidirectional", fallback=False)
    d["sync_batch_norm"] = cp.getboolean(_TRAINING, "sync_batch_norm", fallback=False)   # DP only; deviation from the reference
    # the decoder behind the per-mini-batch training error rate: greedy (GPU) | beam (default: the reference's width-100 beam
    # decoder, models/AcousticModel.py:312-314,:641, on host threads, reported `train_decoder_lag` mini-batches late; 0 = wait)
    d["train_decoder"] = cp.get(_TRAINING, "train_decoder", fallback="beam")
    if d["train_decoder"] not in ("greedy", "beam"):
        raise ValueError("train_decoder must be 'greedy' or 'beam', not %r" % d["train_decoder"])
    d["train_decoder_lag"] = cp.getint(_TRAINING, "train_decoder_lag", fallback=1)
    return d


class HyperParameterHandler(object):
    def __init__(self, config_file):
        hp = self.hyper_params = read_config_file(config_file)
        if hp["log_file"] is not None:
            logging.basicConfig(filename=hp["log_file"])
        logging.getLogger().setLevel(hp["log_level"])
        logging.info("Using checkpoint %s", hp["checkpoint_dir"])
        os.makedirs(hp["checkpoint_dir"], exist_ok=True)
        self.file_path = os.path.join(hp["checkpoint_dir"], "hyperparams.p")
        if not self.check_exists():
            self.save_params(hp)
            logging.info("No hyper params detected at checkpoint... reading config file")
        elif not self.check_changed(hp):
            logging.info("No hyper parameter changed detected, using old checkpoint...")
        elif not hp["use_config_file_if_checkpoint_exists"]:
            self.hyper_params = self.get_params()
            logging.info("Restoring hyper params from previous checkpoint...")
        else:   # structural change + "use config file": start a fresh, timestamped checkpoint dir
            sub = "{0}_hidden_size_{1}_numlayers_{2}_signal_processing_{3}".format(
                int(time.time()), hp["hidden_size"], hp["num_layers"], hp["signal_processing"])
            hp["checkpoint_dir"] = os.path.join(hp["checkpoint_dir"], sub)
            os.makedirs(hp["checkpoint_dir"])
            self.file_path = os.path.join(hp["checkpoint_dir"], "hyperparams.p")
            self.save_params(hp)

    def get_hyper_params(self):
        return self.hyper_params

    def save_params(self, dic):
        with open(self.file_path, "wb") as fh:
            pickle.dump(dic, fh)

    def get_params(self):
        with open(self.file_path, "rb") as fh:
            return pickle.load(fh)

    def check_exists(self):
        return os.path.exists(self.file_path)

    def check_changed(self, new_params):
        if not self.check_exists():
            return False
        old = self.get_params()
        for k, v in _STRUCTURAL_DEFAULTS.items():     # compatibility defaults (older pickles, the reference's own)
            old.setdefault(k, v)
        return any(old[k] != new_params.get(k, _STRUCTURAL_DEFAULTS.get(k)) for k in _STRUCTURAL)

    read_config_file = staticmethod(read_config_file)
