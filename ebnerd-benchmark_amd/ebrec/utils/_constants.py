"""Column names of the EB-NeRD parquet files used on the NRMS path (subset of the reference's
utils/_constants.py:1-52; same names and values so frames are interchangeable)."""

# behaviors.parquet
DEFAULT_USER_COL = "user_id"
DEFAULT_IMPRESSION_ID_COL = "impression_id"
DEFAULT_IMPRESSION_TIMESTAMP_COL = "impression_time"
DEFAULT_INVIEW_ARTICLES_COL = "article_ids_inview"
DEFAULT_CLICKED_ARTICLES_COL = "article_ids_clicked"
DEFAULT_IS_BEYOND_ACCURACY_COL = "is_beyond_accuracy"
DEFAULT_ARTICLE_ID_COL = "article_id"
DEFAULT_SESSION_ID_COL = "session_id"

# articles.parquet
DEFAULT_TITLE_COL = "title"
DEFAULT_SUBTITLE_COL = "subtitle"
DEFAULT_BODY_COL = "body"
DEFAULT_CATEGORY_COL = "category"

# history.parquet
DEFAULT_HISTORY_ARTICLE_ID_COL = f"{DEFAULT_ARTICLE_ID_COL}_fixed"
DEFAULT_HISTORY_IMPRESSION_TIMESTAMP_COL = f"{DEFAULT_IMPRESSION_TIMESTAMP_COL}_fixed"

# created by the pipeline
DEFAULT_LABELS_COL = "labels"
DEFAULT_KNOWN_USER_COL = "is_known_user"
