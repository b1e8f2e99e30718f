"""Host utilities on the NRMS path (reference: utils/_python.py): lookup objects of the loaders,
ranking and the submission file."""
from __future__ import annotations

import json
import zipfile
from pathlib import Path
from typing import Iterable

import numpy as np


def create_lookup_dict(df, key: str, value: str) -> dict:
    """{df[key][i]: df[value][i]} (reference _python.py:397-416)."""
    from ._frames import to_pandas

    df = to_pandas(df)
    return dict(zip(df[key].tolist(), df[value].tolist()))


def create_lookup_objects(lookup_dictionary: dict, unknown_representation: str):
    """(index dict, matrix) for gather-style lookups (reference _python.py:467-484).

    Row i (1-based, dict insertion order) of the matrix is the value of the i-th key; row 0 is the
    "unknown" representation: zeros, or the column mean.  The reference wraps each index in a length-1
    polars Series (a polars ``replace`` speed trick, which is where the singleton axis its loaders squeeze
    comes from); here the index is a plain int.
    """
    lookup_indexes = {key: i for i, key in enumerate(lookup_dictionary, start=1)}
    lookup_matrix = np.array(list(lookup_dictionary.values()))
    if unknown_representation == "zeros":
        unknown = np.zeros(lookup_matrix.shape[1], dtype=lookup_matrix.dtype)
    elif unknown_representation == "mean":
        unknown = np.mean(lookup_matrix, axis=0, dtype=lookup_matrix.dtype)
    else:
        raise ValueError(f"'{unknown_representation}' is not a specified method. Can be either 'zeros' or 'mean'.")
    return lookup_indexes, np.vstack([unknown, lookup_matrix])


def repeat_by_list_values_from_matrix(input_array, matrix: np.ndarray, repeats) -> np.ndarray:
    """matrix[input_array] with row i repeated repeats[i] times (reference _python.py:370-388)."""
    return np.repeat(matrix[np.asarray(input_array)], repeats=np.asarray(repeats), axis=0)


def rank_predictions_by_score(arr: Iterable[float]) -> np.ndarray:
    """1 for the highest score, 2 for the next, ... (reference _python.py:41-59)."""
    return np.argsort(np.argsort(arr)[::-1]) + 1


def write_submission_file(impression_ids: Iterable[int], prediction_scores: Iterable, path=Path("predictions.txt"),
                          rm_file: bool = True, filename_zip: str = None) -> None:
    """One ``<impression_id> [r1,r2,...]`` line per impression, then zipped (reference _python.py:62-90)."""
    path = Path(path)
    with open(path, "w") as f:
        for impr_index, preds in zip(impression_ids, prediction_scores):
            f.write(f"{impr_index} [" + ",".join(str(i) for i in preds) + "]\n")
    zip_submission_file(path=path, rm_file=rm_file, filename_zip=filename_zip)


def read_submission_file(path) -> tuple[list[int], list]:
    impression_ids, prediction_scores = [], []
    with open(path, "r") as file:
        for line in file:
            impid, ranks = line.strip("\n").split()
            impression_ids.append(int(impid))
            prediction_scores.append(json.loads(ranks))
    return impression_ids, prediction_scores


def zip_submission_file(path, filename_zip: str = None, verbose: bool = True, rm_file: bool = True) -> None:
    path = Path(path)
    path_zip = path.parent.joinpath(filename_zip) if filename_zip else path.with_suffix(".zip")
    if path_zip.suffix != ".zip":
        raise ValueError(f"suffix for {path_zip.name} has to be '.zip'")
    if verbose:
        print(f"Zipping {path} to {path_zip}")
    with zipfile.ZipFile(path_zip, "w", zipfile.ZIP_DEFLATED) as f:
        f.write(path, arcname=path.name)
    if rm_file:
        path.unlink()


def write_json_file(dictionary: dict, path) -> None:
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "w") as file:
        json.dump(dictionary, file)
