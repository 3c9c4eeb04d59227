# coding=utf-8
"""Command line of the speech recogniser -- same flags and loop semantics as the reference's
stt.py (argument parser :360-404, train loop + LR plateau rule :171-236, file / evaluate modes
:239-324), driving the MI355X-native AcousticModel instead of a TensorFlow session.

Not kept (out of the hot-path scope): --train_language / --generate_text (the
reference's language model is an unfinished stub), --record (pyaudio), --XLA (no tracing
compiler here; accepted and ignored), TensorBoard.  `training_dataset_dirs` takes the reference's corpus
trees (LibriSpeech / TED-LIUM / Shtooka / Vystadial, walked by rnn_speech_amd.corpus) or a `path<TAB>transcript`
manifest.

Data parallel: `python -m torch.distributed.run --nproc-per-node N stt.py --train_acoustic` -- every rank trains on
its own equal-size shard, gradients are summed over RCCL once per optimiser step, the logged loss / error rate (and
therefore the learning-rate plateau rule) are job-wide means, rank 0 writes the checkpoints.
"""
import argparse
import logging
import os
import sys
from random import shuffle

import numpy as np

from models.AcousticModel import AcousticModel, Session, bucketed_order
from models.SpeechRecognizer import SpeechRecognizer
import util.audioprocessor as audioprocessor
import util.dataprocessor as dataprocessor
import util.hyperparams as hyperparams
from rnn_speech_amd import dataparallel


def main():
    prog_params = parse_args()
    grp = dataparallel.Group.single()
    if prog_params["train_acoustic"]:
        # one process per GPU (torch.distributed.run): joins the job described by RANK / WORLD_SIZE / LOCAL_RANK
        grp = dataparallel.current()
    # rank 0 owns the checkpoint directory (hyperparams.p, a possibly timestamped sub-directory): the others
    # take its view instead of racing it
    hyper_params = None
    if grp.rank == 0:
        hyper_params = hyperparams.HyperParameterHandler(prog_params["config_file"]).get_hyper_params()
    hyper_params = grp.broadcast_object(hyper_params)
    audio_processor = audioprocessor.AudioProcessor(hyper_params["max_input_seq_length"],
                                                    hyper_params["signal_processing"],
                                                    n_mfcc=hyper_params.get("n_mfcc", 20),
                                                    load_sr=hyper_params.get("sample_rate", 22050))
    hyper_params["input_dim"] = audio_processor.feature_size
    speech_reco = SpeechRecognizer(hyper_params["language"])
    hyper_params["char_map"] = speech_reco.get_char_map()
    hyper_params["char_map_length"] = speech_reco.get_char_map_length()

    if prog_params["train_acoustic"]:
        # ONE shuffle / train-test split for the whole job (load_acoustic_dataset shuffles with an unseeded RNG):
        # rank 0 draws it and every rank shards the same permutation -- disjoint shards of equal size, no test
        # item leaking into another rank's training shard
        sets = None
        if grp.rank == 0:
            sets = speech_reco.load_acoustic_dataset(
                hyper_params["training_dataset_dirs"], hyper_params["test_dataset_dirs"],
                hyper_params["training_filelist_cache"],
                hyper_params["dataset_size_ordering"] in ("True", "First_run_only"), hyper_params["train_frac"])
        train_set, test_set = grp.broadcast_object(sets)
        if hyper_params["dataset_size_ordering"] == "Bucketed" and grp.world > 1:
            # GLOBAL length buckets, dealt across the ranks: optimiser step k is the same bucket on every rank, so the all-reduce
            # never waits for a rank that drew a longer mini-batch (bucketing each rank's shard on its own -- rounds 2 - 4 -- made
            # the step the maximum over `world` unrelated buckets).  The shuffle of the buckets is the job's, drawn by rank 0.
            seed = grp.broadcast_object(int.from_bytes(os.urandom(4), "little") if grp.rank == 0 else None)
            train_set = dataparallel.shard_bucketed(train_set, hyper_params["batch_size"], grp.rank, grp.world, seed)
            hyper_params["dataset_size_ordering"] = "Bucketed_by_job"      # (build_acoustic_training_rnn keeps the order)
        else:
            train_set = dataparallel.shard(train_set, grp.rank, grp.world)
        train_acoustic_rnn(train_set, test_set, hyper_params, prog_params)
    elif prog_params["file"] is not None:
        process_file(audio_processor, hyper_params, prog_params["file"])
    elif prog_params["evaluate"]:
        evaluate(hyper_params)
    else:
        sys.exit("mode not supported by the MI355X build (see module docstring)")


def build_acoustic_training_rnn(sess, hyper_params, prog_params, train_set, test_set):
    model = AcousticModel(hyper_params["num_layers"], hyper_params["hidden_size"], hyper_params["batch_size"],
                          hyper_params["max_input_seq_length"], hyper_params["max_target_seq_length"],
                          hyper_params["input_dim"], hyper_params["batch_normalization"],
                          hyper_params["char_map_length"])
    ds_args = (hyper_params["batch_size"], hyper_params["max_input_seq_length"],
               hyper_params["max_target_seq_length"], hyper_params["signal_processing"], hyper_params["char_map"])
    model.precision = hyper_params.get("precision", "f32")
    model.bidirectional = hyper_params.get("bidirectional", False)
    model.sync_batch_norm = hyper_params.get("sync_batch_norm", False)
    model.train_decoder = hyper_params.get("train_decoder", "beam")
    model.train_decoder_lag = hyper_params.get("train_decoder_lag", 1)
    if hyper_params["dataset_size_ordering"] == "Bucketed":
        train_set[:] = bucketed_order(train_set, hyper_params["batch_size"])
    pipe = dict(n_mfcc=hyper_params.get("n_mfcc", 20), prefetch=hyper_params.get("prefetch_batches", 2),
                feature_cache_mb=hyper_params.get("feature_cache_mb", 0),
                sample_rate=hyper_params.get("sample_rate", 22050))
    train_dataset = model.build_dataset(train_set, *ds_args, **pipe)
    test_dataset = model.build_dataset(test_set, *ds_args, **pipe)
    t_iterator, v_iterator = model.add_datasets_input(train_dataset, test_dataset)
    sess.run(t_iterator.initializer)
    sess.run(v_iterator.initializer)
    model.create_training_rnn(hyper_params["dropout_input_keep_prob"], hyper_params["dropout_output_keep_prob"],
                              hyper_params["grad_clip"], hyper_params["learning_rate"],
                              hyper_params["lr_decay_factor"], use_iterator=True)
    model.add_tensorboard(sess, hyper_params["tensorboard_dir"], prog_params["tb_name"], prog_params["timeline"])
    model.initialize(sess)
    model.restore(sess, hyper_params["checkpoint_dir"] + "/acoustic/")
    if prog_params["learn_rate"] is not None:
        model.set_learning_rate(sess, prog_params["learn_rate"])
    return model, t_iterator, v_iterator


class PlateauSchedule(object):
    """The reference's learning-rate rule (stt.py:219-231): keep the mean training error rate of each
    checkpoint window; a new minimum restarts the history, 7 windows without one trigger a decay."""
    PATIENCE = 7

    def __init__(self):
        self.history = []

    def should_decay(self, window_error_rate):
        best_so_far = min(self.history, default=sys.maxsize)
        if window_error_rate <= best_so_far:
            self.history = []
        self.history.append(window_error_rate)
        if len(self.history) < self.PATIENCE:
            return False
        self.history = []
        return True


def _rebuild_training_input(model, sess, iterator, train_set, hp):
    """End of an epoch: reshuffle (unless the corpus is kept size-ordered) and rewind the iterator."""
    order = hp["dataset_size_ordering"]
    if order in ("False", "First_run_only", "Bucketed", "Bucketed_by_job"):
        if order == "Bucketed_by_job":    # data parallel: the job's global buckets in a new order, the same on every rank
            grp = dataparallel.current()
            seed = grp.broadcast_object(int.from_bytes(os.urandom(4), "little") if grp.rank == 0 else None)
            logging.info("Re-drawing the order of the job's length buckets (seed %d)", seed)
            train_set[:] = dataparallel.reshuffle_buckets(train_set, hp["batch_size"], seed)
        elif order == "Bucketed":         # extra mode of this build: similar lengths per batch, batches shuffled
            logging.info("Re-drawing the order of the length-bucketed mini-batches")
            train_set[:] = bucketed_order(train_set, hp["batch_size"])
        else:
            logging.info("Shuffling the training dataset")
            shuffle(train_set)
        sess.run(iterator.make_initializer(iterator.dataset.with_items(train_set)))   # keeps the feature cache
    else:
        logging.info("Reuse the same training dataset")
        sess.run(iterator.initializer)


def train_acoustic_rnn(train_set, test_set, hyper_params, prog_params):
    hp = hyper_params
    ckpt_dir = hp["checkpoint_dir"] + "/acoustic/"
    epoch_limit = prog_params["max_epoch"]
    window = hp["steps_per_checkpoint"]
    with Session() as sess:
        model, t_iterator, v_iterator = build_acoustic_training_rnn(sess, hp, prog_params, train_set, test_set)
        try:
            _train_loop(model, sess, t_iterator, v_iterator, train_set, test_set, hp, prog_params, ckpt_dir, epoch_limit, window)
        finally:
            model.close()       # the asynchronous decoder's threads and pinned buffers: not left to __del__ at interpreter shutdown


def _train_loop(model, sess, t_iterator, v_iterator, train_set, test_set, hp, prog_params, ckpt_dir, epoch_limit, window):
    """The reference's training loop (stt.py:171-236): windows of steps_per_checkpoint optimiser steps, checkpoint, evaluation, plateau rule."""
    schedule = PlateauSchedule()
    epoch = step = 0

    def out_of_epochs():
        return epoch_limit is not None and epoch > epoch_limit

    while not out_of_epochs():
        window_error = 0.0
        for _ in range(window):
            _loss, err, step, exhausted = model.run_train_step(sess, hp["mini_batch_size"],
                                                               hp["rnn_state_reset_ratio"])
            window_error += err / window
            if exhausted:
                epoch += 1
                logging.info("End of epoch number : %d", epoch)
                if out_of_epochs():
                    logging.info("Max number of epochs reached, exiting train step")
                    break
                _rebuild_training_input(model, sess, t_iterator, train_set, hp)
        model.save(sess, ckpt_dir)
        if step % hp["steps_per_evaluation"] == 0 and len(test_set) > 0:
            model.run_evaluation(sess)
            sess.run(v_iterator.initializer)
        if schedule.should_decay(window_error):
            sess.run(model.learning_rate_decay_op)
            logging.info("Model is not improving, decaying the learning rate")
            if model.learning_rate_var.eval() < 1e-7:
                logging.info("Learning rate is too low, exiting")
                return
            model.save(sess, ckpt_dir)      # keep the decayed rate in the checkpoint
    logging.info("Max number of epochs reached, exiting training session")


def _forward_model(hyper_params, batch_size):
    model = AcousticModel(hyper_params["num_layers"], hyper_params["hidden_size"], batch_size,
                          hyper_params["max_input_seq_length"], hyper_params["max_target_seq_length"],
                          hyper_params["input_dim"], hyper_params["batch_normalization"],
                          hyper_params["char_map_length"])
    model.precision = hyper_params.get("precision", "f32")
    model.bidirectional = hyper_params.get("bidirectional", False)
    model.sync_batch_norm = hyper_params.get("sync_batch_norm", False)
    model.create_forward_rnn()
    model.initialize(None)
    model.restore(None, hyper_params["checkpoint_dir"] + "/acoustic/")
    return model


def process_file(audio_processor, hyper_params, file):
    feat_vec, original_length = audio_processor.process_audio_file(file)
    T = hyper_params["max_input_seq_length"]
    if original_length > T:
        logging.warning("File too long: %d frames, truncated to %d", original_length, T)
    padded = np.zeros((T, 1, feat_vec.shape[1]), np.float32)
    padded[:len(feat_vec), 0] = feat_vec
    model = _forward_model(hyper_params, 1)
    predictions = model.process_input(None, padded, [min(original_length, T)])
    transcribed_text = [dataprocessor.DataProcessor.get_labels_str(hyper_params["char_map"], p) for p in predictions]
    print(transcribed_text)
    return transcribed_text


def evaluate(hyper_params):
    if hyper_params["test_dataset_dirs"] is None:
        logging.fatal("Setting test_dataset_dirs in config file is mandatory for evaluation mode")
        sys.exit(1)
    _, test_set = SpeechRecognizer.load_acoustic_dataset(hyper_params["test_dataset_dirs"],
                                                         hyper_params["test_dataset_dirs"])
    if not test_set:
        logging.fatal("No files in test set during an evaluation mode")
        sys.exit(1)
    logging.info("Using %d size of test set", len(test_set))
    model = _forward_model(hyper_params, hyper_params["batch_size"])
    try:
        wer, cer = model.evaluate_full(None, test_set, hyper_params["max_input_seq_length"],
                                       hyper_params["signal_processing"], hyper_params["char_map"],
                                       n_mfcc=hyper_params.get("n_mfcc", 20),
                                       sample_rate=hyper_params.get("sample_rate", 22050))
    finally:
        model.close()
    print("Resulting WER : {0:.3g} %".format(wer))
    print("Resulting CER : {0:.3g} %".format(cer))
    return wer, cer


_MODES = (("train_acoustic", "store_true", "train the acoustic model"),
          ("train_language", "store_true", "reference stub -- not supported here"),
          ("file", str, "transcribe one wav file"),
          ("record", "store_true", "live microphone mode -- not supported here"),
          ("evaluate", "store_true", "WER / CER of the restored model on the test manifest"),
          ("generate_text", "store_true", "reference stub -- not supported here"))


def parse_args():
    """Same option names as the reference (stt.py:360-404); exactly one mode is required."""
    parser = argparse.ArgumentParser(description="MI355X-native rnn-speech command line")
    modes = parser.add_mutually_exclusive_group(required=True)
    for name, kind, text in _MODES:
        if kind == "store_true":
            modes.add_argument("--" + name, action="store_true", default=False, help=text)
        else:
            modes.add_argument("--" + name, type=kind, default=None, help=text)
    parser.add_argument("--config", default="config.ini", help="configuration file (same keys as the reference)")
    parser.add_argument("--tb_name", default=None, help="kept for compatibility (no TensorBoard here)")
    parser.add_argument("--max_epoch", type=int, default=None, help="stop after this many epochs")
    parser.add_argument("--learn_rate", type=float, default=None, help="override the stored learning rate")
    parser.add_argument("--timeline", action="store_true", help="log per-step timings")
    parser.add_argument("--XLA", action="store_true", help="kept for compatibility, ignored")
    ns = parser.parse_args()
    out = {name: getattr(ns, name) for name, _, _ in _MODES}
    out.update(config_file=ns.config, tb_name=ns.tb_name, max_epoch=ns.max_epoch, learn_rate=ns.learn_rate,
               timeline=ns.timeline, XLA=ns.XLA)
    return out


if __name__ == "__main__":
    main()
