"""Real-tokenizer text front-end against the strings the reference builds (ref demo/infer.py:49-58,
134-157, 175).  No tokenizer files exist offline, so a byte-level BPE tokenizer with the Qwen2-VL special tokens is built on the
fly (tokenizers library) and saved as a checkpoint-style directory."""
import numpy as np
import pytest


SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>", "<|image_pad|>", "<|video_pad|>"]


def make_tokenizer_dir(path, with_template=False):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    corpus = ["You are a helpful assistant.", "system user assistant", "Time=0.0-3.0s Time=12.5-13.5s", "Please describe the video.",
              "what is happening now? ...", "livecc 0123456789 =-."] * 4
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=400, special_tokens=SPECIALS,
                                                        initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>",
                                   additional_special_tokens=SPECIALS[1:])
    if with_template:
        fast.chat_template = "{% for m in messages %}<|im_start|>{{ m['role'] }}\n{{ m['content'] }}<|im_end|>\n{% endfor %}"
    fast.save_pretrained(str(path))
    return str(path)


@pytest.fixture(scope="module")
def tokdir(tmp_path_factory):
    return make_tokenizer_dir(tmp_path_factory.mktemp("tok"))


SYS = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"


def test_turn_strings_follow_the_reference(tokdir):
    from livecc_amd.text import TextFrontEnd
    fe = TextFrontEnd(tokdir)
    assert fe.system_prompt_offset == len(SYS)                                    # ref infer.py:51-58
    assert fe.streaming_eos_token_id == fe.tokenizer(" ...").input_ids[-1]        # ref infer.py:49
    first = fe.turn_text(0.0, 3.0, "Please describe the video.", continuing=False)
    assert first == (SYS + "<|im_start|>user\nTime=0.0-3.0s<|vision_start|><|video_pad|><|vision_end|>Please describe the video."
                     "<|im_end|>\n<|im_start|>assistant\n")
    later = fe.turn_text(3.0, 4.0, None, continuing=True)
    assert later == ("<|im_end|>\n<|im_start|>user\nTime=3.0-4.0s<|vision_start|><|video_pad|><|vision_end|><|im_end|>\n"
                     "<|im_start|>assistant\n")                                   # ref infer.py:149-150
    changed = fe.turn_text(4.0, 5.0, "what is happening now?", continuing=True)
    assert changed.endswith("<|vision_end|>what is happening now?<|im_end|>\n<|im_start|>assistant\n")
    qa = fe.qa_text("what is happening now?", continuing=False, with_video=True)
    assert qa == SYS + "<|im_start|>user\n<|vision_start|><|video_pad|><|vision_end|>what is happening now?<|im_end|>\n<|im_start|>assistant\n"
    assert fe.qa_text("and then?", continuing=True, with_video=False) == "<|im_end|>\n<|im_start|>user\nand then?<|im_end|>\n<|im_start|>assistant\n"


def test_placeholder_expansion_and_round_trip(tokdir):
    from livecc_amd.text import TextFrontEnd
    fe = TextFrontEnd(tokdir)
    vid = fe.video_token_id
    ids = fe.turn_ids(0.0, 3.0, (3, 28, 52), "Please describe the video.", continuing=False)
    assert int((ids == vid).sum()) == 3 * 28 * 52 // 4                            # HF processing_qwen2_vl.py:63-66
    runs = np.flatnonzero(np.diff((ids == vid).astype(np.int8)))
    assert len(runs) == 2, "the placeholders form one contiguous run"
    tok = fe.tokenizer
    assert ids[runs[0]] == tok.convert_tokens_to_ids("<|vision_start|>") and ids[runs[1] + 1] == tok.convert_tokens_to_ids("<|vision_end|>")
    text = fe.decode(ids, skip_special_tokens=False)
    assert text == fe.turn_text(0.0, 3.0, "Please describe the video.", False).replace("<|video_pad|>", "<|video_pad|>" * (3 * 28 * 52 // 4))
    assert fe.decode(fe.tokenizer("what is happening now?").input_ids) == "what is happening now?"
    with pytest.raises(ValueError):
        fe.encode("no placeholder here", [(1, 4, 4)])
    # a text-only chat template in the tokenizer cannot render video items -> the Qwen2-VL restatement is used
    import tempfile
    fe2 = TextFrontEnd(make_tokenizer_dir(tempfile.mkdtemp(), with_template=True))
    assert "<|video_pad|>" in fe2.turn_text(0.0, 3.0, None, False)


def test_special_ids_are_checked_against_the_model_config(tokdir):
    import dataclasses
    from livecc_amd.config import tiny
    from livecc_amd.text import TextFrontEnd
    with pytest.raises(ValueError):
        TextFrontEnd(tokdir, tiny())                                              # tiny() uses other special ids
    fe = TextFrontEnd(tokdir)
    cfg = dataclasses.replace(tiny(), video_token_id=fe.video_token_id, eos_token_id=fe.eos_token_id)
    assert TextFrontEnd(tokdir, cfg).merge == cfg.spatial_merge_size
