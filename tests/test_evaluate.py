"""Offline data-parallel commentary driver (ref evaluation/livesports3kcc/distributed_generate_livecc.py): prompt text,
strided sharding, resume, JSONL merge -- host logic, run with a stand-in for the model."""
import json
import os

from livecc_amd import evaluate as E


def test_prompts_follow_the_reference():
    assert E.overall_prompt("Final", "he shoots", simple_ctx=True) == "he shoots"                 # title OR preasr
    assert E.overall_prompt("Final", "", simple_ctx=True) == "Final"
    p = E.overall_prompt("Final", "he shoots")
    assert p == (E.COMMENTARY_PROMPT + 'This is a video titled "Final".\nHere is previous commentary of the video:\n\nhe shoots\n\n'
                 "Please continue to comment the video.")
    assert E.overall_prompt(None, None) == E.COMMENTARY_PROMPT
    assert E.merge_commentary([(0, 3, "A pass ..."), (3, 4, ""), (4, 5, " ..."), (5, 6, "and a goal")]) == "A pass  and a goal..."


class FakeInfer:
    def __init__(self):
        self.calls = []

    def live_cc_once_for_evaluation(self, query, video, video_start=0, video_end=None, remote_loader=None, max_new_tokens=32,
                                    repetition_penalty=1.05):
        # the reference's signature (ref demo/infer.py:245-253)
        self.calls.append((video, query, video_start, max_new_tokens, repetition_penalty))
        return [[video_start, video_start + 3, f"clip{video}"], [video_start + 3, video_start + 4, " ..."]]


def test_strided_sharding_resume_and_merge(tmp_path):
    recs = [dict(frames=(lambda i=i: i), video_id=f"v{i}", event_id=i, begin=float(i), end=float(i) + 4, event_title="t", preasr_text="")
            for i in range(7)]
    save = str(tmp_path / "out" / "model")
    f0, f1 = FakeInfer(), FakeInfer()
    assert E.generate_shard(recs, f0, save, rank=0, world=2) == 4          # 0, 2, 4, 6
    assert [c[0] for c in f0.calls] == [0, 2, 4, 6] and f0.calls[0][3:] == (32, 1.15)
    os.remove(os.path.join(save, "4.json"))                                  # a lost item is redone, finished ones are skipped
    assert E.generate_shard(recs, f0, save, rank=0, world=2) == 1 and f0.calls[-1][0] == 4
    assert E.generate_shard(recs, f1, save, rank=1, world=2) == 3          # 1, 3, 5
    path = E.merge_jsonl(save)
    lines = [json.loads(l) for l in open(path)]
    assert [l["event_id"] for l in lines] == list(range(7)) and not os.path.exists(save)
    assert lines[3] == {"video_id": "v3", "event_id": 3, "begin": 3.0, "end": 7.0, "pred": "clip3..."}
