# coding=utf-8
"""Command line of the speech recogniser -- same flags and loop semantics as the reference's
stt.py (argument parser :360-404, train loop + LR plateau rule :171-236, file / evaluate modes
:239-324), driving the MI355X-native AcousticModel instead of a TensorFlow session.

Not kept (out of the hot-path scope): --train_language / --generate_text (the
reference's language model is an unfinished stub), --record (pyaudio), --XLA (no tracing
compiler here; accepted and ignored), TensorBoard.  Dataset discovery takes a JSON/TSV manifest
(`path<TAB>transcript` per line) in `training_dataset_dirs` instead of walking corpus trees.
"""
import argparse
import logging
import os
import sys
from random import shuffle

import numpy as np

from models.AcousticModel import AcousticModel, Session
from models.SpeechRecognizer import SpeechRecognizer
import util.audioprocessor as audioprocessor
import util.dataprocessor as dataprocessor
import util.hyperparams as hyperparams


def load_manifest(paths):
    """`a.tsv, b.tsv` -> [[audio_path, cleaned transcript, None], ...]"""
    items = []
    for p in [q.strip() for q in (paths or "").split(",") if q.strip()]:
        with open(p) as fh:
            for line in fh:
                if "\t" in line:
                    wav, text = line.rstrip("\n").split("\t", 1)
                    items.append([wav, dataprocessor.DataProcessor.clean_label(text), None])
    return items


def init_distributed():
    """One process per GPU (torchrun / torch.distributed.run): RCCL through the 'nccl' backend."""
    import torch
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group("nccl")
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def main():
    prog_params = parse_args()
    serializer = hyperparams.HyperParameterHandler(prog_params["config_file"])
    hyper_params = serializer.get_hyper_params()
    audio_processor = audioprocessor.AudioProcessor(hyper_params["max_input_seq_length"],
                                                    hyper_params["signal_processing"],
                                                    n_mfcc=hyper_params.get("n_mfcc", 20))
    hyper_params["input_dim"] = audio_processor.feature_size
    speech_reco = SpeechRecognizer(hyper_params["language"])
    hyper_params["char_map"] = speech_reco.get_char_map()
    hyper_params["char_map_length"] = speech_reco.get_char_map_length()

    if prog_params["train_acoustic"]:
        rank, world = init_distributed()
        train_set = load_manifest(hyper_params["training_dataset_dirs"])
        test_set = load_manifest(hyper_params["test_dataset_dirs"])
        if hyper_params["dataset_size_ordering"] not in ("True", "First_run_only"):
            shuffle(train_set)
        train_set = train_set[rank::world]          # data parallel: shard utterances by rank
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
    train_dataset = model.build_dataset(train_set, *ds_args, n_mfcc=hyper_params.get("n_mfcc", 20))
    test_dataset = model.build_dataset(test_set, *ds_args, n_mfcc=hyper_params.get("n_mfcc", 20))
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


def train_acoustic_rnn(train_set, test_set, hyper_params, prog_params):
    with Session() as sess:
        model, t_iterator, v_iterator = build_acoustic_training_rnn(sess, hyper_params, prog_params,
                                                                    train_set, test_set)
        previous_mean_error_rates = []
        current_step = epoch = 0
        while True:
            mean_error_rate = 0
            for _ in range(hyper_params["steps_per_checkpoint"]):
                _loss, step_err, current_step, dataset_empty = model.run_train_step(
                    sess, hyper_params["mini_batch_size"], hyper_params["rnn_state_reset_ratio"])
                mean_error_rate += step_err / hyper_params["steps_per_checkpoint"]
                if dataset_empty:
                    epoch += 1
                    logging.info("End of epoch number : %d", epoch)
                    if prog_params["max_epoch"] is not None and epoch > prog_params["max_epoch"]:
                        logging.info("Max number of epochs reached, exiting train step")
                        break
                    if hyper_params["dataset_size_ordering"] in ("False", "First_run_only"):
                        logging.info("Shuffling the training dataset")
                        shuffle(train_set)
                        train_dataset = model.build_dataset(
                            train_set, hyper_params["batch_size"], hyper_params["max_input_seq_length"],
                            hyper_params["max_target_seq_length"], hyper_params["signal_processing"],
                            hyper_params["char_map"], n_mfcc=hyper_params.get("n_mfcc", 20))
                        sess.run(t_iterator.make_initializer(train_dataset))
                    else:
                        sess.run(t_iterator.initializer)
            model.save(sess, hyper_params["checkpoint_dir"] + "/acoustic/")
            if current_step % hyper_params["steps_per_evaluation"] == 0 and len(test_set) > 0:
                model.run_evaluation(sess)
                sess.run(v_iterator.initializer)
            # plateau rule: 7 checkpoint windows without a new minimum -> decay the learning rate
            if mean_error_rate <= min(previous_mean_error_rates, default=sys.maxsize):
                previous_mean_error_rates.clear()
            previous_mean_error_rates.append(mean_error_rate)
            if len(previous_mean_error_rates) >= 7:
                sess.run(model.learning_rate_decay_op)
                previous_mean_error_rates.clear()
                logging.info("Model is not improving, decaying the learning rate")
                if model.learning_rate_var.eval() < 1e-7:
                    logging.info("Learning rate is too low, exiting")
                    break
                model.save(sess, hyper_params["checkpoint_dir"] + "/acoustic/")
            if prog_params["max_epoch"] is not None and epoch > prog_params["max_epoch"]:
                logging.info("Max number of epochs reached, exiting training session")
                break


def _forward_model(hyper_params, batch_size):
    model = AcousticModel(hyper_params["num_layers"], hyper_params["hidden_size"], batch_size,
                          hyper_params["max_input_seq_length"], hyper_params["max_target_seq_length"],
                          hyper_params["input_dim"], hyper_params["batch_normalization"],
                          hyper_params["char_map_length"])
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
    test_set = load_manifest(hyper_params["test_dataset_dirs"])
    if not test_set:
        logging.fatal("Missing test_dataset_dirs in config file")
        sys.exit(1)
    logging.info("Using %d size of test set", len(test_set))
    model = _forward_model(hyper_params, hyper_params["batch_size"])
    wer, cer = model.evaluate_full(None, test_set, hyper_params["max_input_seq_length"],
                                   hyper_params["signal_processing"], hyper_params["char_map"],
                                   n_mfcc=hyper_params.get("n_mfcc", 20))
    print("Resulting WER : {0:.3g} %".format(wer))
    print("Resulting CER : {0:.3g} %".format(cer))
    return wer, cer


def parse_args():
    parser = argparse.ArgumentParser()
    parser.set_defaults(train_acoustic=False, train_language=False, file=None, record=False, evaluate=False,
                        generate_text=False)
    group = parser.add_mutually_exclusive_group(required=True)
    group.add_argument("--train_acoustic", dest="train_acoustic", action="store_true",
                       help="Train the acoustic network")
    group.add_argument("--train_language", dest="train_language", action="store_true",
                       help="(reference stub; not supported)")
    group.add_argument("--file", type=str, help="Path to a wav file to process")
    group.add_argument("--record", dest="record", action="store_true", help="(not supported)")
    group.add_argument("--evaluate", dest="evaluate", action="store_true", help="Evaluate WER against the test_set")
    group.add_argument("--generate_text", dest="generate_text", action="store_true", help="(not supported)")
    parser.add_argument("--XLA", dest="XLA", action="store_true", help="accepted for compatibility, ignored")
    parser.add_argument("--timeline", dest="timeline", action="store_true", help="log per-step timings")
    parser.add_argument("--config", type=str, default="config.ini", help="Path to configuration file.")
    parser.add_argument("--tb_name", type=str, default=None, help="accepted for compatibility")
    parser.add_argument("--max_epoch", type=int, default=None, help="Max epoch to train (no limitation if not provided)")
    parser.add_argument("--learn_rate", type=float, default=None, help="Force learning rate to a specific value")
    args = parser.parse_args()
    return {"config_file": args.config, "train_acoustic": args.train_acoustic, "train_language": args.train_language,
            "file": args.file, "record": args.record, "evaluate": args.evaluate, "generate_text": args.generate_text,
            "XLA": args.XLA, "timeline": args.timeline, "tb_name": args.tb_name, "max_epoch": args.max_epoch,
            "learn_rate": args.learn_rate}


if __name__ == "__main__":
    main()
