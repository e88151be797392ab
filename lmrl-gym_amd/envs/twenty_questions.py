"""Twenty Questions — dual-model rollouts: a policy asks yes/no questions about a hidden object, a SECOND resident
model (the oracle) answers them inside the rollout loop.

Counterpart of llm_rl_scripts/twenty_questions/env/{env,data,oracle}.py:
  * `TwentyQuestionsPolicyEnvironment` / `BatchedTwentyQuestionsPolicyEnvironment` (env.py:9-141): reset draws the hidden word
    (`random.Random(seed).choice(word_list)`, or `seed % len(word_list)` in deterministic mode) and returns `(Text("Questions:\\n"),)`;
    step sends the stripped question to the oracle, appends `Text(answer + "\\n")`, reward -1 per question (0 for the question that
    guesses the word), done when the word is guessed or after `max_conversation_length` questions (data.py:83-116).
  * `TwentyQuestionsOracle.generate_answers(words, questions)` protocol and the prompt / answer post-processing of `T5Oracle`
    (oracle.py:20-87): prompt "Answer the question about the object truthfully. ...", greedy generation of <= 4 tokens, first
    `yes|no` match -> "Yes." / "No.", anything else -> "No.", INVALID_QUESTION -> "No." without consulting the model output.
  * `asker_postproc*`, `is_done`, `WordVariants`, the default object list (data.py:20-81,292-391).

Two things the reference takes from third parties that are absent here (no network, not vendored):
  * the oracle is flan-T5-XL (JaxSeq T5): `ModelOracle` below accepts ANY batched text generator (`generate(prompts) -> completions`);
    `GPT2EngineOracle` runs it on this package's HIP engine with a second GPT-2-family model resident next to the policy's;
  * `is_done` relies on NLTK's perceptron POS tagger (`nltk.pos_tag(nltk.word_tokenize(...))`).  The tagger is pluggable
    (`set_pos_tagger`); the default uses nltk when importable and otherwise a documented rule-based stand-in.  tests/golden/
    twenty_questions.json pins everything else against the reference run with the SAME stand-in tagger injected.
"""
from __future__ import annotations

import random
import re
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

from ..environment import BatchedTextEnv, StepResult, Text, TextEnv, TextHistory, TextTrajectory

INVALID_QUESTION = "Is this a valid question?\n"
INITIAL_STR = "Questions:\n"

PosTagger = Callable[[str], List[Tuple[str, str]]]

_DETERMINERS = {"a", "an", "the", "this", "that", "these", "those", "it", "its", "your", "my", "some", "any"}
_FUNCTION_WORDS = {"is", "are", "does", "do", "can", "could", "would", "will", "was", "were", "be", "been", "has", "have", "had", "of", "in", "on", "at",
                   "to", "for", "with", "by", "from", "or", "and", "not", "you", "i", "we", "they", "he", "she", "than", "as", "if", "used", "made",
                   "found", "use", "make", "eat", "wear", "play", "hold", "see", "bigger", "smaller", "larger", "alive", "living", "edible", "big",
                   "small", "large", "heavy", "light", "red", "green", "blue", "yellow", "round", "soft", "hard", "electronic", "wooden", "metal"}


def rule_pos_tag(text: str) -> List[Tuple[str, str]]:
    """Stand-in for `nltk.pos_tag(nltk.word_tokenize(text))` when nltk is absent: tokens = runs of letters/digits/apostrophes or single
    punctuation marks; determiners 'DT', a closed list of function / common non-noun words 'XX', punctuation as itself, digits 'CD',
    everything else 'NN' (plural -s 'NNS').  Only the `NN` prefix and the token strings matter to `is_done`."""
    toks = re.findall(r"[A-Za-z0-9']+|[^\sA-Za-z0-9']", text)
    out = []
    for t in toks:
        if not t[0].isalnum():
            out.append((t, t))
        elif t.isdigit():
            out.append((t, "CD"))
        elif t in _DETERMINERS:
            out.append((t, "DT"))
        elif t in _FUNCTION_WORDS:
            out.append((t, "XX"))
        else:
            out.append((t, "NNS" if t.endswith("s") and len(t) > 3 else "NN"))
    return out


def _nltk_pos_tag(text: str) -> List[Tuple[str, str]]:
    import nltk
    return nltk.pos_tag(nltk.word_tokenize(text))


_pos_tagger: Optional[PosTagger] = None


def set_pos_tagger(fn: Optional[PosTagger]) -> None:
    """Inject the POS tagger (`text -> [(token, tag)]`); None restores the default (nltk if importable, else `rule_pos_tag`)."""
    global _pos_tagger
    _pos_tagger = fn


def pos_tag(text: str) -> List[Tuple[str, str]]:
    global _pos_tagger
    if _pos_tagger is None:
        try:
            import nltk  # noqa: F401
            _pos_tagger = _nltk_pos_tag
        except Exception:
            _pos_tagger = rule_pos_tag
    return _pos_tagger(text)


@dataclass
class WordVariants:
    """data.py:20-49: the spellings of one object ("Pants;Pant;Pair of pants") and their POS-tagged token lists."""
    words: List[str]
    pos_tags: List[List[Tuple[str, str]]]

    @classmethod
    def from_list(cls, words_list: List[str]) -> "WordVariants":
        return cls(words=list(words_list), pos_tags=[pos_tag(w.lower()) for w in words_list])

    @classmethod
    def from_str(cls, words_str: str) -> "WordVariants":
        return cls.from_list(words_str.split(";"))

    def __len__(self):
        return len(self.words)

    def __getitem__(self, idx):
        assert 0 <= idx < len(self.words), f"Index {idx} out of range"
        return self.words[idx]

    def json(self):
        return self.words.copy()

    def __str__(self):
        return f"({', '.join(self.words)})"

    def __repr__(self) -> str:
        return f"WordVariants([{', '.join(self.words)}])"


# The task's object list (data, as the Wordle vocabulary files are): llm_rl_scripts/twenty_questions/env/data.py:52-70
DEFAULT_OBJECT_DICT: Dict[str, List[str]] = {
    "Sports": ["Basketball", "Football", "Baseball", "Soccer ball", "Golf ball", "Tennis ball", "Volleyball", "Tennis racket", "Baseball bat", "Helmet"],
    "Animals": ["Cat", "Dog", "Horse", "Cow", "Sheep", "Rabbit", "Lion", "Tiger", "Bear", "Elephant"],
    "Fruits": ["Apple", "Banana", "Orange", "Strawberry", "Grape", "Watermelon", "Pineapple", "Mango", "Cantaloupe", "Peach"],
    "Vehicles": ["Car", "Truck", "Motorcycle", "Boat", "Airplane;Plane", "Train", "Bus", "Helicopter", "Scooter", "Ship"],
    "Clothes": ["Shirt", "Pants;Pant;Pair of pants", "Jacket", "Dress", "Skirt", "Belt", "Shoes;Shoe;Pair of shoes", "Boots;Boot;Pair of boots",
                "Socks;Sock;Pair of socks", "Hat", "Scarf"],
    "Electronics": ["Computer", "Smartphone", "Television;TV", "Headphone;Headphones;Pair of headphones", "Monitor;Computer monitor", "Camera",
                    "Microwave;Microwave oven", "Refrigerator", "Blender", "Computer keyboard;Keyboard"],
    "Musical Instruments": ["Piano", "Guitar", "Drum;Drums", "Violin", "Saxophone", "Flute", "Trumpet", "Clarinet", "Harp", "Trombone"],
    "Furniture": ["Chair", "Table", "Bed", "Desk", "Couch", "Dresser", "Bookcase", "Nightstand", "Mattress", "Pillow"],
    "Office Supplies": ["Pen", "Paper;Piece of paper", "Stapler", "Printer", "Calculator", "Battery;Battery pack;Pack of batteries", "Toothbrush",
                        "Toothpaste", "Pencil", "Sharpie", "Scissors;Pair of scissors", "Key", "Diary", "Calendar"],
    "Vegetables": ["Carrot", "Potato", "Broccoli", "Tomato", "Onion", "Spinach", "Corn", "Peas;Pea", "Celery", "Cucumber"],
    "Art": ["Painting;Canvas painting;Oil painting;Watercolor painting", "Paintbrush", "Canvas;Painting canvas", "Eraser;Pencil eraser", "Marker",
            "Glue;Glue stick;Bottle of glue", "Sculpture"],
    "Kitchen Tools": ["Knife", "Spoon", "Fork", "Plate", "Bowl", "Cooking pot;Pot", "Pan;Saucepan;Frying pan", "Cup",
                      "Chopstick;Chopsticks;Pair of chopsticks", "Whisk"],
    "Nature": ["Rock", "Tree", "Bush", "Mountain", "Forest", "Ocean", "Sea", "Lake", "River", "Meteorite", "Cactus"],
    "Toys": ["Lego;Lego set", "Doll;Toy doll;Plush doll", "Kite", "Puzzle;Jigsaw puzzle", "Stuffed animal"],
    "Jewelry": ["Earring;Earrings;Pair of earrings", "Necklace", "Bracelet", "Ring", "Brooch", "Hairclip", "Pendant", "Watch", "Locket"],
    "Garden Supplies": ["Gloves;Glove;Pair of gloves", "Shovel", "Rake", "Watering can", "Lawn mower"],
    "Tools": ["Hammer", "Screwdriver", "Wrench", "Saw", "Pliers;plier;Pair of pliers", "Drill"],
}


def get_default_word_list() -> List[WordVariants]:
    return [WordVariants.from_str(w) for words in DEFAULT_OBJECT_DICT.values() for w in words]


# ---------------------------------------------------------------------------------------------- trajectory / done logic
def is_done(word_var: WordVariants, question: str) -> bool:
    """data.py:351-391: the question names the hidden object — no noun other than the object's own tokens (and the generic
    'object / something / type / kind', and counter nouns followed by 'of') appears, and one spelling's tokens END the question."""
    while len(question) > 0 and not question[-1].isalpha():
        question = question[:-1]
    if len(question) == 0:
        return False
    qpos = pos_tag(question.lower())
    ignores = {"object", "something", "type", "kind"}
    for plist in word_var.pos_tags:
        ignores.update(w for w, _ in plist)
    for i, (w, tag) in enumerate(qpos):
        if tag[:2] == "NN" and w not in ignores:
            if i + 1 < len(qpos) and qpos[i + 1][0] == "of":
                continue
            return False
    for wpos in word_var.pos_tags:
        if len(wpos) > len(qpos):
            continue
        if all(vw == qw for (vw, _), (qw, _) in zip(wpos, qpos[-len(wpos):])):
            return True
    return False


def create_trajectory_from_history(word_var: WordVariants, text_history: TextHistory, max_conversation_len: int = 20) -> TextTrajectory:
    """data.py:83-116."""
    assert len(text_history) % 2 == 1, "TextHistory should be [initial str, question1, answer1, ..., questionN, answerN]."
    assert all(t.is_action for t in text_history[1::2]), "All questions should be actions."
    assert all(not t.is_action for t in text_history[0::2]), "All answers should not be actions."
    conversation_len = (len(text_history) - 1) // 2
    assert conversation_len <= max_conversation_len, f"Conversation is too long {conversation_len}. Max should be {max_conversation_len}."
    reward = [-1.0 if t.is_action else 0.0 for t in text_history]
    if len(text_history) < 2:
        done = False
    else:
        last_question, last_answer = text_history[-2].text.strip(), text_history[-1].text.strip()
        word_guessed = last_answer == "Yes." and is_done(word_var, last_question)
        done = word_guessed or conversation_len == max_conversation_len
        if word_guessed:
            reward[-2] = 0.0
    return TextTrajectory(tuple(text_history), tuple(reward), done)


def asker_postproc(question: str) -> str:
    """data.py:292-315: normalise a generated question; anything that is not a yes/no question becomes INVALID_QUESTION."""
    question = question.strip()
    if len(question) == 0:
        return INVALID_QUESTION
    if question[-1] != "?":
        question += "?"
    question = question[0].upper() + question[1:]
    if len(question.split(" ")) > 40:
        return INVALID_QUESTION
    if question.split(" ")[0] not in ["Is", "Does", "Can", "Do", "Are", "Could"]:
        return INVALID_QUESTION
    if question[-2] == "." and question.split(" ")[-1] != "etc.?":
        return INVALID_QUESTION
    return question + "\n"


def asker_postproc_simple(question: str) -> str:
    """data.py:318-329."""
    question = question.strip()
    if len(question) == 0:
        return "?\n"
    if question[-1] != "?":
        question += "?"
    return question + "\n"


def asker_postproc_filter_repeats(question: str) -> str:
    """data.py:332-348."""
    question = question.strip()
    if len(question) == 0:
        return "?\n"
    words = question.split(" ")
    if len(words) > 50:
        question = " ".join(words[:50])
    if question[-1] != "?":
        question += "?"
    return question + "\n"


# ---------------------------------------------------------------------------------------------- oracle
class TwentyQuestionsOracle:
    """oracle.py:14-17."""

    def generate_answers(self, words, questions, return_full: bool = False):
        raise NotImplementedError


def get_oracle_prompt(word: WordVariants, question: str) -> str:
    """oracle.py:20-28 (`get_t5_oracle_prompt`)."""
    return ("Answer the question about the object truthfully.\n"
            f"object: {word}\n"
            f"question: {question}\n"
            "answer (yes or no): ")


_ANSWER_RE = re.compile(r"(yes|no)")


def answers_from_outputs(questions: Sequence[str], output_strs: Sequence[str]) -> Tuple[List[str], List[str]]:
    """oracle.py:62-79: INVALID_QUESTION -> "No."; else strip + lower, `re.match(r"(yes|no)")` -> capitalised + "." ; no match -> "No."."""
    answers, full = [], []
    for q, out in zip(questions, output_strs):
        if q == INVALID_QUESTION:
            answers.append("No."); full.append("No.")
            continue
        a_full = out.strip().lower()
        mt = _ANSWER_RE.match(a_full)
        answers.append(mt[0].capitalize() + "." if mt is not None else "No.")
        full.append(a_full)
    return answers, full


class ModelOracle(TwentyQuestionsOracle):
    """`T5Oracle.generate_answers` (oracle.py:44-87) over any batched text generator: `generate(prompts: List[str]) -> List[str]` must
    return the greedy completion (the reference: do_sample=False, max_new_tokens=4, eos = '\\n')."""

    def __init__(self, generate: Callable[[List[str]], List[str]]):
        self.generate = generate

    def generate_answers(self, words: Union[WordVariants, List[WordVariants]], questions: Union[str, List[str]], return_full: bool = False):
        input_is_list = isinstance(words, list)
        if not input_is_list:
            assert not isinstance(questions, list)
            words, questions = [words], [questions]
        assert len(words) == len(questions)
        outs = self.generate([get_oracle_prompt(w, q) for w, q in zip(words, questions)])
        answers, full = answers_from_outputs(questions, outs)
        if not input_is_list:
            answers, full = answers[0], full[0]
        return (answers, full) if return_full else answers


class GPT2EngineOracle(ModelOracle):
    """The oracle as a SECOND model resident on the GPU next to the policy's: greedy generation of <= `max_new_tokens` tokens from the
    oracle prompt on this package's HIP engine (`GPT2PPOPolicy` machinery: chunked prefill, KV cache, fused LM-head sampling).  The
    reference's oracle model is flan-T5-XL (3rd-party JaxSeq T5, not in scope of the engine): any GPT-2-family checkpoint fine-tuned
    on the same (prompt -> yes/no) data drops in here."""

    def __init__(self, engine, tokenizer, max_input_length: int = 124, max_new_tokens: int = 4, eos_token_id: Optional[int] = None):
        from ..policies import GPT2PPOPolicy
        self._policy = GPT2PPOPolicy(engine, tokenizer, max_input_length=max_input_length, max_new_tokens=max_new_tokens, do_sample=False,
                                     eos_token_id=eos_token_id)

        def generate(prompts: List[str]) -> List[str]:
            hist = [(Text(p, False),) for p in prompts]
            out = self._policy.act(hist, [False] * len(hist))
            return [h[-1].text for h in out]
        super().__init__(generate)


# ---------------------------------------------------------------------------------------------- environments
class TwentyQuestionsPolicyEnvironment(TextEnv):
    """env.py:9-64."""

    def __init__(self, oracle: TwentyQuestionsOracle, word_list: List[WordVariants], max_conversation_length: int = 20):
        self.oracle, self.word_list, self.max_conversation_length = oracle, word_list, max_conversation_length
        self.random = random.Random(None)
        self.count = 0
        self.curr_word: Optional[WordVariants] = None

    def step(self, text_history: TextHistory) -> Tuple[TextHistory, float, bool]:
        assert text_history[-1].is_action
        assert self.curr_word is not None, "call env.reset() first."
        self.count += 1
        question = text_history[-1].text.strip()
        answer = self.oracle.generate_answers(self.curr_word, question)
        traj = create_trajectory_from_history(self.curr_word, tuple(text_history) + (Text(answer + "\n", is_action=False),), self.max_conversation_length)
        return traj.text_history, traj.reward[-2], traj.done

    def reset(self, seed: Optional[int] = None, options: Optional[Dict] = None) -> TextHistory:
        self.count = 0
        if seed is not None:
            self.random = random.Random(seed)
        options = options or {}
        if options.get("deterministic", False):
            assert seed is not None, "In deterministic mode, the seed specifies which word to use."
            self.curr_word = self.word_list[seed % len(self.word_list)]
        else:
            self.curr_word = self.random.choice(self.word_list)
        return (Text(INITIAL_STR, is_action=False),)

    def copy(self):
        return TwentyQuestionsPolicyEnvironment(self.oracle, self.word_list, self.max_conversation_length)


class BatchedTwentyQuestionsPolicyEnvironment(BatchedTextEnv):
    """env.py:67-141: ONE oracle call answers the questions of all live slots of the batch (padding slots ask INVALID_QUESTION about
    word_list[0], as the reference does)."""

    def __init__(self, oracle: TwentyQuestionsOracle, word_list: List[WordVariants], max_conversation_length: int = 20, bsize: Optional[int] = None):
        self.bsize, self.oracle, self.word_list, self.max_conversation_length = bsize, oracle, word_list, max_conversation_length
        self.randoms = [random.Random(None) for _ in range(bsize or 0)]
        self.curr_words: Optional[List[WordVariants]] = None

    def step(self, text_history_batch: List[Optional[TextHistory]], done: Optional[List[bool]] = None,
             done_batch: Optional[List[bool]] = None) -> List[Optional[StepResult]]:
        # (`interact_environment` passes `done=` (LLM_RL/environment.py:186); the reference's signature names it `done_batch`, which makes
        #  its own batched env unusable under its own driver — both spellings are accepted here; finished slots arrive as None either way)
        assert self.curr_words is not None, "call env.reset() first."
        if self.bsize is None:
            self.bsize = len(text_history_batch)
        npad = self.bsize - len(text_history_batch)
        questions = [h[-1].text.strip() if h is not None else INVALID_QUESTION for h in text_history_batch]
        answers = self.oracle.generate_answers(self.curr_words + [self.word_list[0]] * npad, questions + [INVALID_QUESTION] * npad)[: self.bsize - npad]
        results: List[Optional[StepResult]] = []
        for answer, word, h in zip(answers, self.curr_words, text_history_batch):
            if h is None:
                results.append(None)
                continue
            traj = create_trajectory_from_history(word, tuple(h) + (Text(answer + "\n", is_action=False),), self.max_conversation_length)
            results.append((traj.text_history, traj.reward[-2], traj.done))
        return results

    def reset(self, seed_batch: Optional[List[Optional[int]]] = None, options_batch: Optional[List[Optional[Dict]]] = None) -> List[TextHistory]:
        if seed_batch is None:
            seed_batch = [None] * self.bsize
        if options_batch is None:
            options_batch = [{} for _ in range(len(seed_batch))]
        self.randoms, self.curr_words = [], []
        out = []
        for i, (seed, options) in enumerate(zip(seed_batch, options_batch)):
            self.randoms.append(random.Random(seed))
            options = options or {}
            if options.get("deterministic", False):
                assert seed is not None, "In deterministic mode, the seed specifies which word to use."
                self.curr_words.append(self.word_list[seed % len(self.word_list)])
            else:
                self.curr_words.append(self.randoms[i].choice(self.word_list))
            out.append((Text(INITIAL_STR, is_action=False),))
        return out

    def copy(self):
        return BatchedTwentyQuestionsPolicyEnvironment(self.oracle, self.word_list, self.max_conversation_length, self.bsize)
