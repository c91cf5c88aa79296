"""Label alphabets (reference: data/label_sets.py:2-13): 27 symbols + CTC blank '_' at
index 0 + space ' ' last = 29 labels."""
english_labels = ["'"] + [chr(ord('A') + i) for i in range(26)]
english_lowercase_labels = [s.lower() for s in english_labels]
hebrew_labels = ['א', 'ב', 'ג', 'ד', 'ה', 'ו', 'ז', 'ח', 'ט', 'י', 'כ', 'ל', 'מ', 'נ', 'ס', 'ע', 'פ', 'צ', 'ק', 'ר', 'ש',
                 'ת', 'ן', 'ף', 'ץ', 'ם', 'ך']

labels_map = {'english': english_labels, 'hebrew': hebrew_labels, 'english_lowercase': english_lowercase_labels}
for _lang in labels_map:
    labels_map[_lang].insert(0, '_')   # CTC blank label; blank index is 0
    labels_map[_lang].append(' ')
