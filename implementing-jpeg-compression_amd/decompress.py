"""Entry script: `python decompress.py out.bin rec.png` (the reference's decompress.py)."""
from cli import decompress, main_decompress  # noqa: F401

if __name__ == "__main__":
    main_decompress()
