"""End-to-end NRMS run on EB-NeRD with the MI355X-native model: the stage order, flag names and artefacts of the
reference's examples/reproducibility_scripts/ebnerd_nrms.py (train + validation -> Wu-2019 sampling -> fit with
EarlyStopping / ModelCheckpoint / ReduceLROnPlateau on val_auc -> chunked test prediction with --chunks_done
resume -> ranked submission zip), on pandas frames and without TensorFlow.

    python examples/reproducibility_scripts/ebnerd_nrms.py --data_path ~/ebnerd_data --datasplit ebnerd_small
    python -m torch.distributed.run --nproc-per-node 8 examples/reproducibility_scripts/ebnerd_nrms.py ...   # data parallel

Deviation from the reference, on purpose: the reference script loads the XLM-R word embeddings but never passes
them to NRMSModel (token ids then overflow its 32000-row Glorot table). Here the
table always covers the tokenizer's id range: the HF word embeddings when the HF model loads from a local cache,
otherwise a Glorot table of --vocab_size rows for the built-in hash tokenizer.
"""
from __future__ import annotations

import datetime as dt
import gc
import os
import shutil
import sys
import zlib
from pathlib import Path

import numpy as np
import pandas as pd

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "ebnerd-benchmark_amd"))
sys.path.insert(0, str(Path(__file__).resolve().parent))

from args_nrms import get_args  # noqa: E402
from ebrec.evaluation import AucScore, MetricEvaluator, MrrScore, NdcgScore  # noqa: E402
from ebrec.models.newsrec import NRMSModel  # noqa: E402
from ebrec.models.newsrec.callbacks import EarlyStopping, ModelCheckpoint, ReduceLROnPlateau, TensorBoard  # noqa: E402
from ebrec.models.newsrec.dataloader import NRMSDataLoader, NRMSDataLoaderPretransform  # noqa: E402
from ebrec.models.newsrec.model_config import hparams_nrms, hparams_to_dict, print_hparams  # noqa: E402
from ebrec.utils._articles import convert_text2encoding_with_transformers, create_article_id_to_value_mapping  # noqa: E402
from ebrec.utils._behaviors import (add_prediction_scores, create_binary_labels_column, ebnerd_from_path,  # noqa: E402
                                    sampling_strategy_wu2019)
from ebrec.utils._constants import (DEFAULT_BODY_COL, DEFAULT_CLICKED_ARTICLES_COL, DEFAULT_HISTORY_ARTICLE_ID_COL,  # noqa: E402
                                    DEFAULT_IMPRESSION_ID_COL, DEFAULT_IMPRESSION_TIMESTAMP_COL, DEFAULT_INVIEW_ARTICLES_COL,
                                    DEFAULT_IS_BEYOND_ACCURACY_COL, DEFAULT_LABELS_COL, DEFAULT_SUBTITLE_COL, DEFAULT_TITLE_COL,
                                    DEFAULT_USER_COL)
from ebrec.utils._frames import concat_str_columns, split_df_chunks, with_column  # noqa: E402
from ebrec.utils._python import rank_predictions_by_score, write_json_file, write_submission_file  # noqa: E402


def hash_tokenize(texts, max_length: int, vocab_size: int):
    """Offline stand-in for the HF tokenizer: whitespace tokens hashed into [1, vocab_size); 0 pads."""
    out = []
    for t in texts:
        ids = [1 + zlib.crc32(w.encode()) % (vocab_size - 1) for w in str(t).lower().split()][:max_length]
        out.append(ids + [0] * (max_length - len(ids)))
    return out


def encode_articles(df_articles, args):
    df_articles, cat_col = concat_str_columns(df_articles, columns=[DEFAULT_TITLE_COL, DEFAULT_SUBTITLE_COL, DEFAULT_BODY_COL])
    word2vec_embedding = None
    if args.tokenizer in ("auto", "hf"):
        try:
            from transformers import AutoModel, AutoTokenizer

            tok = AutoTokenizer.from_pretrained(args.transformer_model_name, local_files_only=True)
            word2vec_embedding = AutoModel.from_pretrained(args.transformer_model_name, local_files_only=True) \
                .get_input_embeddings().weight.detach().numpy()
            df_articles, token_col = convert_text2encoding_with_transformers(df_articles, tok, cat_col, max_length=args.max_title_length)
            return create_article_id_to_value_mapping(df_articles, value_col=token_col), word2vec_embedding, len(tok)
        except Exception as e:  # no network / no local cache
            if args.tokenizer == "hf":
                raise
            print(f"HF tokenizer unavailable ({type(e).__name__}); using the hash tokenizer")
    df_articles = with_column(df_articles, "tokens", hash_tokenize(df_articles[cat_col], args.max_title_length, args.vocab_size))
    return create_article_id_to_value_mapping(df_articles, value_col="tokens"), None, args.vocab_size


def main(argv=None):
    args = get_args(argv)
    hparams = hparams_nrms
    hparams.title_size, hparams.history_size = args.max_title_length, args.history_size
    hparams.head_num, hparams.head_dim, hparams.attention_hidden_dim = args.head_num, args.head_dim, args.attention_hidden_dim
    hparams.optimizer, hparams.loss, hparams.dropout, hparams.learning_rate = args.optimizer, args.loss, args.dropout, args.learning_rate
    hparams.newsencoder_units_per_layer = None
    print("Initiating articles...")
    PATH = Path(args.data_path).expanduser()
    article_mapping, word2vec_embedding, vocab = encode_articles(pd.read_parquet(PATH / "articles.parquet"), args)

    def build_model():
        model = NRMSModel(hparams=hparams, word2vec_embedding=word2vec_embedding, word_emb_dim=args.word_emb_dim, vocab_size=vocab,
                          seed=42, train_embedding=not args.freeze_embedding, shard_table=args.shard_table, precision=args.precision)
        model._engine.enable_graphs(not args.no_graph)
        return model

    return run(args, hparams, build_model, article_mapping, NRMSModel.__name__)


def run(args, hparams, build_model, article_mapping, MODEL_NAME, after_validation=None):
    """Everything after the article representation is built: shared by the NRMS and NRMSDocVec drivers.
    after_validation(model, ctx): when given, it runs after training + the held-out-day metrics INSTEAD of the test-set
    prediction, and its result is returned next to the history and the metrics (the history-length sweep of
    ebnerd_nrms_doc_hist.py ends that way: reference lines 250-300 have no test-set part)."""
    import torch

    rank, world = 0, int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group("nccl")
        rank = torch.distributed.get_rank()
    for arg, val in vars(args).items():
        print(f"{arg} : {val}")
    PATH = Path(args.data_path).expanduser()
    SEED, DATASPLIT = args.seed, args.datasplit
    Loader = NRMSDataLoaderPretransform if args.nrms_loader == "NRMSDataLoaderPretransform" else NRMSDataLoader
    print_hparams(hparams)

    DUMP_DIR = Path(args.dump_dir)
    MODEL_OUTPUT_NAME = f"{MODEL_NAME}-{dt.datetime.now():%Y%m%d-%H%M%S}"
    if world > 1:  # one run = one artefact directory: rank 0's time stamp names it on every rank
        name = [MODEL_OUTPUT_NAME]
        torch.distributed.broadcast_object_list(name, src=0)
        MODEL_OUTPUT_NAME = name[0]
    ARTIFACT_DIR = DUMP_DIR / "test_predictions" / MODEL_OUTPUT_NAME
    MODEL_WEIGHTS = DUMP_DIR / f"state_dict/{MODEL_OUTPUT_NAME}/weights"
    LOG_DIR = DUMP_DIR / f"runs/{MODEL_OUTPUT_NAME}"
    TEST_CHUNKS_DIR = ARTIFACT_DIR / "test_chunks"
    if rank == 0:
        TEST_CHUNKS_DIR.mkdir(parents=True, exist_ok=True)
        MODEL_WEIGHTS.parent.mkdir(parents=True, exist_ok=True)
    if world > 1:
        torch.distributed.barrier()
    COLUMNS = [DEFAULT_IMPRESSION_TIMESTAMP_COL, DEFAULT_HISTORY_ARTICLE_ID_COL, DEFAULT_INVIEW_ARTICLES_COL,
               DEFAULT_CLICKED_ARTICLES_COL, DEFAULT_IMPRESSION_ID_COL, DEFAULT_USER_COL]
    if rank == 0:
        write_json_file(hparams_to_dict(hparams), ARTIFACT_DIR / f"{MODEL_NAME}_hparams.json")
        write_json_file(vars(args), ARTIFACT_DIR / f"{MODEL_NAME}_argparser.json")

    # train + validation together, last day held out (reference lines 158-188)
    df = pd.concat([ebnerd_from_path(PATH / DATASPLIT / "train", history_size=args.history_size, padding=0),
                    ebnerd_from_path(PATH / DATASPLIT / "validation", history_size=args.history_size, padding=0)], ignore_index=True)
    df = df[df[DEFAULT_HISTORY_ARTICLE_ID_COL].notna()]
    df = df.sample(frac=args.train_fraction, random_state=SEED)[COLUMNS]
    df = sampling_strategy_wu2019(df, npratio=args.npratio, shuffle=True, with_replacement=True, seed=SEED)
    df = create_binary_labels_column(df)
    if world > 1:
        # data parallel: every rank trains and validates on its own slice of the impressions.  fit() runs the same
        # number of steps on every rank (the shortest shard decides) and all-reduces the epoch logs, so the callbacks
        # below (early stopping, LR schedule, best-weights restore) take identical decisions on every rank.
        df = df.iloc[rank::world]
    days = pd.to_datetime(df[DEFAULT_IMPRESSION_TIMESTAMP_COL]).dt.date
    last_dt = days.max() - dt.timedelta(days=1)
    df_train, df_validation = df[days < last_dt], df[days >= last_dt]
    print(f"train rows {len(df_train)}, validation rows {len(df_validation)}")
    mk = lambda frame, eval_mode, bs: Loader(behaviors=frame, article_dict=article_mapping, unknown_representation="zeros",
                                             history_column=DEFAULT_HISTORY_ARTICLE_ID_COL, eval_mode=eval_mode, batch_size=bs)
    train_dataloader, val_dataloader = mk(df_train, False, args.bs_train), mk(df_validation, False, args.bs_train)

    callbacks = [TensorBoard(log_dir=LOG_DIR, histogram_freq=1),
                 EarlyStopping(monitor="val_auc", mode="max", patience=4, restore_best_weights=True),
                 ModelCheckpoint(filepath=str(MODEL_WEIGHTS), monitor="val_auc", mode="max", save_best_only=True,
                                 save_weights_only=True, verbose=1),
                 ReduceLROnPlateau(monitor="val_auc", mode="max", factor=0.2, patience=2, min_lr=1e-6)]
    model = build_model()
    model.model.compile(optimizer=model.model.optimizer, loss=model.model.loss, metrics=["AUC"])
    model.model.summary()
    hist = model.model.fit(train_dataloader, validation_data=val_dataloader, epochs=args.epochs, callbacks=callbacks)
    if MODEL_WEIGHTS.exists():
        print(f"loading model: {MODEL_WEIGHTS}")
        model.model.load_weights(str(MODEL_WEIGHTS))

    # offline metrics on the held-out day with the per-impression evaluator (quick-start nrms_ebnerd.py:250-262)
    df_val_eval = df_validation.reset_index(drop=True)
    pred_val = model.scorer.predict(mk(df_val_eval, True, args.bs_test))
    df_val_eval = add_prediction_scores(df_val_eval, pred_val.tolist())
    labels, scores = df_val_eval[DEFAULT_LABELS_COL].tolist(), df_val_eval["scores"].tolist()
    if world > 1:  # each rank scored its own shard of the held-out day; the metrics are over all of it, on every rank
        parts = [None] * world
        torch.distributed.all_gather_object(parts, ([list(l) for l in labels], [list(x) for x in scores]))
        labels, scores = [l for p in parts for l in p[0]], [x for p in parts for x in p[1]]
    metrics = MetricEvaluator(labels=labels, predictions=scores,
                              metric_functions=[AucScore(), MrrScore(), NdcgScore(k=5), NdcgScore(k=10)]).evaluate()
    if rank == 0:
        print(metrics)
        write_json_file(metrics.evaluations, ARTIFACT_DIR / "validation_metrics.json")

    if after_validation is not None:
        ctx = dict(args=args, PATH=PATH, DATASPLIT=DATASPLIT, SEED=SEED, COLUMNS=COLUMNS, ARTIFACT_DIR=ARTIFACT_DIR,
                   article_mapping=article_mapping, rank=rank, world=world)
        return hist, metrics.evaluations, after_validation(model, ctx)

    # Test prediction is rank 0's job.  With a row-sharded table every lookup is a collective, so the other ranks run
    # the same frames alongside (identical call sequence) and simply do not write; with a replicated table they leave.
    sharded = bool(getattr(args, "shard_table", False)) and world > 1
    if rank != 0 and not sharded:
        return hist, metrics.evaluations
    writer = rank == 0
    # ---- test set: fake labels, BA split, chunked prediction with resume (reference lines 263-365)
    print("Initiating testset...")
    df_test = ebnerd_from_path(PATH / "ebnerd_testset" / "test", history_size=args.history_size, padding=0)
    df_test = df_test[df_test[DEFAULT_HISTORY_ARTICLE_ID_COL].notna()].sample(frac=args.fraction_test, random_state=SEED)
    df_test = with_column(df_test, DEFAULT_CLICKED_ARTICLES_COL, [[l[0]] for l in df_test[DEFAULT_INVIEW_ARTICLES_COL]])
    df_test = df_test[COLUMNS + [DEFAULT_IS_BEYOND_ACCURACY_COL]]
    df_test = with_column(df_test, DEFAULT_LABELS_COL, [[0] * len(l) for l in df_test[DEFAULT_INVIEW_ARTICLES_COL]])
    ba = df_test[DEFAULT_IS_BEYOND_ACCURACY_COL].astype(bool)
    df_wo, df_w = df_test[~ba], df_test[ba]

    def predict_frame(frame, bs):
        scores = model.scorer.predict(mk(frame, True, bs))
        frame = add_prediction_scores(frame, scores.tolist())
        return with_column(frame, "ranked_scores", [list(rank_predictions_by_score(x)) for x in frame["scores"]])

    # --chunks_done N resumes a crashed run: finished chunks are re-read from <dump_dir>/resume (the reference
    # keeps them under its time-stamped artefact directory, where a restarted run cannot find them)
    RESUME_DIR = Path(args.dump_dir) / "resume"
    if writer:
        RESUME_DIR.mkdir(parents=True, exist_ok=True)
    if sharded:
        torch.distributed.barrier()
    chunks = split_df_chunks(df_wo, n_chunks=args.n_chunks_test)
    done = [pd.read_parquet(RESUME_DIR / f"pred_wo_ba_{i}.parquet") for i in range(1, args.chunks_done + 1)]
    for i, chunk in enumerate(chunks[args.chunks_done:], start=1 + args.chunks_done):
        print(f"Test chunk: {i}/{len(chunks)}")
        chunk = predict_frame(chunk, args.batch_size_test_wo_b)[[DEFAULT_IMPRESSION_ID_COL, "ranked_scores"]]
        if writer:
            chunk.to_parquet(TEST_CHUNKS_DIR / f"pred_wo_ba_{i}.parquet")
            chunk.to_parquet(RESUME_DIR / f"pred_wo_ba_{i}.parquet")
        done.append(chunk)
        gc.collect()
    print("Initiating testset with beyond-accuracy...")
    pred_w = predict_frame(df_w, args.batch_size_test_w_b)[[DEFAULT_IMPRESSION_ID_COL, "ranked_scores"]] if len(df_w) else None
    if not writer:
        return hist, metrics.evaluations
    df_out = pd.concat(done + ([pred_w] if pred_w is not None else []), ignore_index=True)
    df_out.to_parquet(ARTIFACT_DIR / "test_predictions.parquet")
    shutil.rmtree(TEST_CHUNKS_DIR, ignore_errors=True)
    shutil.rmtree(RESUME_DIR, ignore_errors=True)
    write_submission_file(impression_ids=df_out[DEFAULT_IMPRESSION_ID_COL], prediction_scores=df_out["ranked_scores"],
                          path=ARTIFACT_DIR / "predictions.txt", filename_zip=f"{MODEL_NAME}-{SEED}-{DATASPLIT}.zip")
    return hist, metrics.evaluations


if __name__ == "__main__":
    main()
