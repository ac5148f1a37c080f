"""Drop-in mirror of the reference's `main_code/utils` package (same module and symbol names) over
the MI355X-native engine in `frx`.  Scripts written against the reference --
`from utils.criterion import ArcFaceNet`, `from utils.model_utils import main_pipeline` -- run
unchanged with this directory's parent on sys.path / as cwd, exactly as `main_code/` is used upstream."""
