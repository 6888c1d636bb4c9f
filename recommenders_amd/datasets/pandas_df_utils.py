"""DataFrame -> libffm text, the step right before the xDeepFM hot path.

Mirror of `recommenders.datasets.pandas_df_utils.LibffmConverter`
(reference recommenders/datasets/pandas_df_utils.py:86-262): same constructor, `fit`,
`transform`, `fit_transform`, `get_params`, same indexing rule and the same output strings
(pinned by the reference's own expected values, tests/test_libffm_converter.py).

Differences, on purpose:
  * the reference's dtype check uses `np.float`, which numpy >= 1.24 no longer has, so its `fit`
    raises AttributeError there; here "float" means any numpy floating dtype;
  * `transform` does not overwrite the columns of the caller's frame (the reference does);
  * the encoding is vectorised (`pandas.factorize` per column) instead of a Python loop over
    every cell; the file is written with one `"\\n".join` instead of `np.savetxt`.
"""
import numpy as np
import pandas as pd

DEFAULT_RATING_COL = "rating"


class LibffmConverter:
    """Converts a dataframe to libffm format `<field_index>:<feature_index>:<value>`.

    Columns of dtype object (strings) are categorical: every distinct value of the column gets its
    own feature index and the value 1.  Numeric columns (int / float) are one feature each, the
    cell is the value.  Indices start at 1 and run over the fields in column order
    (reference pandas_df_utils.py:204-219).

    Args:
        filepath (str): path to save the converted data (optional).

    Attributes:
        field_count (int): count of fields in the libffm data
        feature_count (int): count of features in the libffm data
        filepath (str or None): where the output is stored
    """

    def __init__(self, filepath=None):
        self.filepath = filepath
        self.col_rating = None
        self.field_names = None
        self.field_count = None
        self.feature_count = None

    def fit(self, df, col_rating=DEFAULT_RATING_COL):
        """Check the input columns (reference :150-180); returns self."""
        types = df.dtypes
        if not all(x == object or np.issubdtype(x, np.integer) or np.issubdtype(x, np.floating) for x in types):
            raise TypeError("Input columns should be only object and/or numeric types.")
        if col_rating not in df.columns:
            raise TypeError("Column of {} is not in input dataframe columns".format(col_rating))
        self.col_rating = col_rating
        self.field_names = list(df.drop(col_rating, axis=1).columns)
        return self

    def transform(self, df):
        """Transform a dataset with the fitted schema to libffm format (reference :182-239)."""
        if self.col_rating not in df.columns:
            raise ValueError(
                "Input dataset does not contain the label column {} in the fitting dataset".format(self.col_rating)
            )
        if not all(x in df.columns for x in self.field_names):
            raise ValueError("Not all columns in the input dataset appear in the fitting dataset")

        idx = 1
        self.field_feature_dict = {}
        out = {self.col_rating: df[self.col_rating].values}
        for col_index, field in enumerate(self.field_names):
            col = df[field]
            prefix = str(col_index + 1) + ":"
            if col.dtype == object:
                # first-appearance order, like the reference's dict insertion
                codes, uniques = pd.factorize(col.values, sort=False)
                for k, feature in enumerate(uniques):
                    self.field_feature_dict[(field, feature)] = idx + k
                feat_idx = (codes + idx).astype(str)
                # a non-string object (e.g. a number stored in an object column) keeps its value
                vals = np.array(["1" if isinstance(v, str) else str(v) for v in col.values], dtype=object)
                cells = prefix + pd.Series(feat_idx, dtype=object) + ":" + pd.Series(vals, dtype=object)
                idx += len(uniques)
            else:
                for feature in pd.unique(col.values):
                    self.field_feature_dict[(field, feature)] = idx
                cells = prefix + str(idx) + ":" + col.map(str).astype(object).reset_index(drop=True)
                idx += 1
            out[field] = cells.values
        self.field_count = len(self.field_names)
        self.feature_count = idx - 1

        result = pd.DataFrame(out, index=df.index, columns=[self.col_rating] + self.field_names)
        if self.filepath is not None:
            rows = result.astype(str).agg(" ".join, axis=1)
            with open(self.filepath, "w") as f:
                f.write("\n".join(rows.values))
                f.write("\n")
        return result

    def fit_transform(self, df, col_rating=DEFAULT_RATING_COL):
        """fit + transform (reference :241-251)."""
        return self.fit(df, col_rating=col_rating).transform(df)

    def get_params(self):
        """Field count, feature count and file path (reference :253-263)."""
        return {"field count": self.field_count, "feature count": self.feature_count, "file path": self.filepath}
