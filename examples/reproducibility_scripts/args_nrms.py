"""Command line of the NRMS / NRMSDocVec drivers: the flag names and defaults of the reference's
examples/reproducibility_scripts/args_nrms.py:4-126 and args_nrms_docvec.py, plus the MI355X-only switches."""
import argparse


def build_parser(docvec: bool = False) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MI355X-native NRMS training / test prediction on EB-NeRD")
    p.add_argument("--data_path", type=str, default="~/ebnerd_data", help="Path to the data directory")
    p.add_argument("--seed", type=int, default=123)
    p.add_argument("--datasplit", type=str, default="ebnerd_small")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--bs_train", type=int, default=32)
    p.add_argument("--bs_test", type=int, default=32)
    p.add_argument("--batch_size_test_wo_b", type=int, default=32)
    p.add_argument("--batch_size_test_w_b", type=int, default=4)
    p.add_argument("--history_size", type=int, default=20)
    p.add_argument("--npratio", type=int, default=4)
    p.add_argument("--epochs", type=int, default=5)
    p.add_argument("--train_fraction", type=float, default=1.0)
    p.add_argument("--fraction_test", type=float, default=1.0)
    p.add_argument("--nrms_loader", type=str, default="NRMSDataLoaderPretransform",
                   choices=["NRMSDataLoaderPretransform", "NRMSDataLoader"])
    p.add_argument("--n_chunks_test", type=int, default=10)
    p.add_argument("--chunks_done", type=int, default=0, help="resume the test prediction after this many chunks")
    p.add_argument("--transformer_model_name", type=str, default="FacebookAI/xlm-roberta-large")
    p.add_argument("--max_title_length", type=int, default=30)
    p.add_argument("--head_num", type=int, default=16 if docvec else 20)
    p.add_argument("--head_dim", type=int, default=16 if docvec else 20)
    p.add_argument("--attention_hidden_dim", type=int, default=200)
    p.add_argument("--optimizer", type=str, default="adam")
    p.add_argument("--loss", type=str, default="cross_entropy_loss")
    p.add_argument("--dropout", type=float, default=0.20)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    if docvec:
        p.add_argument("--document_embeddings", type=str, default="document_vector.parquet",
                       help="parquet with article_id + a vector column (relative to --data_path)")
        p.add_argument("--title_size", type=int, default=768)
        p.add_argument("--newsencoder_units_per_layer", nargs="+", type=int, default=[512, 512, 512])
        p.add_argument("--newsencoder_l2_regularization", type=float, default=1e-4)
    # --- not in the reference -------------------------------------------------------------------
    p.add_argument("--dump_dir", type=str, default="ebnerd_predictions")
    p.add_argument("--tokenizer", type=str, default="auto", choices=["auto", "hf", "hash"],
                   help="'hf' = AutoTokenizer(transformer_model_name) (needs a local HF cache), 'hash' = whitespace "
                        "tokens hashed into --vocab_size ids, 'auto' = hf if it loads else hash")
    p.add_argument("--vocab_size", type=int, default=32000, help="embedding rows with the hash tokenizer")
    p.add_argument("--word_emb_dim", type=int, default=300)
    p.add_argument("--freeze_embedding", action="store_true", help="frozen lookup table (BASELINE config 2)")
    p.add_argument("--shard_table", action="store_true", help="row-shard the embedding table over the ranks")
    p.add_argument("--no_graph", action="store_true", help="do not capture the train step into hipGraphs")
    p.add_argument("--precision", type=str, default="exact", choices=["exact", "split"],
                   help="exact: every matmul on the exact-fp32 MFMA kernels; split: the news encoder's projection GEMMs as fp32-accurate "
                        "bf16x6 split products on the bf16 matrix pipe (NRMS only)")
    p.add_argument("--filter_min_history", type=int, default=100,
                   help="ebnerd_nrms_doc_hist.py: keep validation users with at least this many history entries "
                        "(the reference hard-codes 100, ebnerd_nrms_doc_hist.py:253)")
    return p


def get_args(argv=None, docvec: bool = False):
    return build_parser(docvec).parse_args(argv)
