"""sage_amd: MI355X-native ablation hot path for SaGe 2.0 behind the reference's Python API
(`sage_tokenizer.{SaGeVocabBuilder, SaGeTokenizer, setSageFolder}`, reference __init__.py:1-3).  See DESIGN.md."""
from .builder import SaGeVocabBuilder
from .folders import setSageFolder
from .tokenizer import SaGeTokenizer

__all__ = ["SaGeVocabBuilder", "SaGeTokenizer", "setSageFolder"]
