"""Dataset-side types the model plugin consumes (openkge/dataset.py:25-39)."""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class EntityRelationDatasetMeta:
    """Same fields as the reference's dataclass; the model reads entities_size, relations_size,
    min_entities_size and min_relations_size (ids 0 and 1 are PAD/UNK, openkge/index_mapper.py:14)."""
    entity_id_count_map: dict = field(default_factory=dict)
    relation_id_count_map: dict = field(default_factory=dict)
    entity_token_id_count_map: dict = field(default_factory=dict)
    relation_token_id_count_map: dict = field(default_factory=dict)
    entity_id_to_tokens_map: dict = field(default_factory=dict)
    relation_id_to_tokens_map: dict = field(default_factory=dict)
    entities_size: int = 0
    relations_size: int = 0
    min_entities_size: int = 2
    min_relations_size: int = 2
    entity_tokens_size: int = 0
    relation_tokens_size: int = 0
    max_length: int = 1
